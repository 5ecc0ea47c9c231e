#!/bin/bash
# does the runtime's scratch policy (use-once above HSA_SCRATCH_SINGLE_LIMIT) cost the spilling kernels a fixed time per launch?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export AB_STEPS=10
for lim in default 1073741824 8589934592; do
  echo "== HSA_SCRATCH_SINGLE_LIMIT=$lim"
  if [ $lim = default ]; then unset HSA_SCRATCH_SINGLE_LIMIT; else export HSA_SCRATCH_SINGLE_LIMIT=$lim; fi
  timeout -s KILL 200 python tools/r06/fork_ab.py "50000 100000 300000 1000000" "AMX_FORK=0" 2>&1 | grep "^AMX"
  AMX_GCERT2_THIRD=1 timeout -s KILL 200 python tools/r05/proto_fit.py 105 300000 5 2>&1 | tail -2 | head -1 | cut -c1-250
done
