#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "200000 1000000" default 2>&1
AMX_TILE_F32=1 bash tools/r04/ab.sh "200000 1000000" default 2>&1 | sed 's/^default/tilef32/'
python tools/r04/skew_ab.py 1000000 2>&1 | grep "voxels "
