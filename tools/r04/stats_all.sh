#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
AMICO_AMD_LIB=$PWD/variants/stats/libamico_amd.so AMX_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 0 --voxels ${1:-1000000} --no-cpu-baseline --no-other-configs 2>&1 | grep "^\[amx\]" | grep -v "no error" | tail -20
