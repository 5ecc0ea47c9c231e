#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "200000 1000000 4000000" default 2>&1
python tools/r04/skew_ab.py 1000000 2>&1 | grep "voxels \|populations"
python tools/r04/skew_ab.py 300000 2>&1 | grep "voxels \|populations"
