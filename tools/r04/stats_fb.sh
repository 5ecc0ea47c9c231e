#!/bin/bash
# how often a stage-1 seed trip repeats its scan by hand because some lane's best atom is a banned one (AMX_STATS build: slot "store" counts those trips)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
AMICO_AMD_LIB=$PWD/variants/stats/libamico_amd.so AMX_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 0 --voxels ${1:-1000000} --no-cpu-baseline --no-other-configs 2>&1 | grep "^\[amx\] \(seeds\|seed solver\|stage-3\)" | tail -6
