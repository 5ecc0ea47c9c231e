#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 300000 1000000; do
python tools/r04/skew_ab.py $n 2>&1 | grep -v amdgpu.ids
AMICO_AMD_LIB=$PWD/variants/nosteal/libamico_amd.so python tools/r04/skew_ab.py $n 2>&1 | grep "voxels "
done
