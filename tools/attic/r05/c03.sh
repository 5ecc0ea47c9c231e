#!/bin/bash
# third session, call 3: jitter of the narrowed float64 call with threads that stay awake through a call (10 calls each)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 300 python tools/r05/host_trace.py 1000000 10 2>&1 | grep "^float" | tee $O/c03_jitter.txt
AMX_HOST_TRACE=1 timeout 300 python tools/r05/host_trace.py 1000000 4 > $O/host_trace2.txt 2>&1
AMX_HOST_THREADS=16 timeout 300 python tools/r05/host_trace.py 1000000 10 2>&1 | grep "^float64 h" | tee -a $O/c03_jitter.txt
AMX_HOST_THREADS=8 timeout 300 python tools/r05/host_trace.py 1000000 10 2>&1 | grep "^float64 h" | tee -a $O/c03_jitter.txt
