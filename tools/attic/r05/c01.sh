#!/bin/bash
# third session, call 1: the float32 transport of float64 host signals (amx_stage.hpp): test, timeline, threads sweep
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -5 | tee $O/c01_tests.txt
AMX_HOST_TRACE=1 timeout 300 python tools/r05/host_trace.py > $O/host_trace1.txt 2>&1; grep -v "batch" $O/host_trace1.txt | tail -12
for t in 4 8 24; do echo "threads $t"; AMX_HOST_THREADS=$t timeout 300 python tools/r05/host_trace.py 2>&1 | grep "^float"; done | tee $O/c01_threads.txt
