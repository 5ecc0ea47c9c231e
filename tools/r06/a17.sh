#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_a17
timeout -s KILL 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_multi.py -m gpu -q 2>&1 | tail -2
for sw in "AMX_HOST_NATIVE32=0" "X=0"; do
  for rep in 1 2; do env $sw timeout -s KILL 200 python tools/r05/host_trace.py 1000000 8 2>&1 | grep "median" | cut -c1-60,100-220 | sed "s/^/$sw: /"; done
done | tee gpurun_out/r06_a17/native32_ab.txt
for pf in 0 1; do AMX_HOST_PREFETCH=$pf timeout -s KILL 200 python tools/r06/first_call_parts.py 2>&1 | grep "PARTS"; done | tee -a gpurun_out/r06_a17/native32_ab.txt
