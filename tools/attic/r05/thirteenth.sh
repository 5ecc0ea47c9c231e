#!/bin/bash
# round 5, thirteenth GPU call: why do the LASSO certificates leave 6.8 % of a 288-volume fit's voxels?  (AMX_STATS build, counters of every pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export AMICO_AMD_LIB=$PWD/variants/stats/libamico_amd.so
for p in hcp bench; do
  AMX_DEBUG=1 python tools/r05/proto_fit.py $p 300000 1 2>&1 | grep "^\[amx\] Gram\|^\[amx\] LASSO\|^\[amx\] seeds\|^hcp\|^bench" | tail -8
done
