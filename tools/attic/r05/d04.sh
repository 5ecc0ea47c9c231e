#!/bin/bash
# fourth session, call 4: results of the finished batches copied home during the solver's tail (default) against one copy at the end
# (AMX_HOST_LATE_RESULTS=1): host_trace.py, 1 M and 400 000 voxels, alternating processes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
for rep in 1 2 3; do for late in 0 1; do for n in 1000000 400000; do
  echo "late_results=$late n=$n"
  if [ $late = 1 ]; then export AMX_HOST_LATE_RESULTS=1; else unset AMX_HOST_LATE_RESULTS; fi
  timeout 300 python tools/r05/host_trace.py $n 8 2>&1 | grep "^float"
done; done; done | tee $O/d04_early_results.txt
unset AMX_HOST_LATE_RESULTS
AMX_HOST_TRACE=1 timeout 300 python tools/r05/host_trace.py 1000000 4 2>&1 | grep -E "^---|last enqueue" | head -8
