#!/bin/bash
# batch plan of the pipelined host-buffer call, re-scanned on the round-6 floor (T(n) = 1.1 ms + 4.8 ns n): fewer, larger batches?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a34; mkdir -p $O
for ramp in 131072 163840 196608 262144; do
for batch in 393216 450000 600000; do
  echo "ramp $ramp batch $batch" | tee -a $O/scan.txt
  AMX_HOST_RAMP=$ramp AMX_HOST_BATCH=$batch timeout -s KILL 200 python tools/r05/host_trace.py 1000000 8 2>&1 | grep "^float" | cut -c1-200 | tee -a $O/scan.txt
done
done
