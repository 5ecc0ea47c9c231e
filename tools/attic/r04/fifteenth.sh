#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default head 2>&1
bash tools/r04/trace1m.sh default 2>&1 | head -16
for v in default head; do
  unset AMICO_AMD_LIB; [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  python tools/r04/skew_ab.py 1000000 2>&1 | grep "voxels "
done
