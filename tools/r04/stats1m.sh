#!/bin/bash
# refusal reasons and phase cycles of the seed solvers / certificates (-DAMX_STATS build in variants/stats), 1 M voxels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
AMICO_AMD_LIB=$PWD/variants/stats/libamico_amd.so AMX_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 0 --voxels ${1:-1000000} --no-cpu-baseline --no-other-configs 2>&1 | grep "^\[amx\] \(seeds\|Gram\|LASSO\|screened\|seed solver\|stage-3\|dual-vector\)" | tail -14
