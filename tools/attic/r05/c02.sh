#!/bin/bash
# third session, call 2: batch plan of the host-buffer calls now that the float64 copy is half as long (ramp x batch)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
for ramp in 32768 65536 131072; do for batch in 262144 393216 480000; do
  echo "ramp $ramp batch $batch"; AMX_HOST_RAMP=$ramp AMX_HOST_BATCH=$batch timeout 300 python tools/r05/host_trace.py 2>&1 | grep "^float"
done; done | tee $O/c02_plan.txt
