#!/bin/bash
# third session, call 5: where the narrowing threads run (AMX_HOST_PIN = gpu | caller | 0), threads asleep between jobs; four processes each
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
for rep in 1 2 3 4; do for pin in gpu caller 0; do
  echo "pin $pin"; AMX_HOST_SPIN_US=0 AMX_HOST_PIN=$pin timeout 300 python tools/r05/host_trace.py 1000000 6 2>&1 | grep "^float64 h"
done; done | tee $O/c05_pin.txt
