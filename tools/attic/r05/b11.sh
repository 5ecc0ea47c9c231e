#!/bin/bash
# a third LASSO certificate pass (supports of 19 .. 24 atoms): 288 volumes (tile read from L2) and the bench shape
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
for t in 0 1; do
  echo "== AMX_GCERT2_THIRD=$t"
  AMX_GCERT2_THIRD=$t python tools/r05/proto_fit.py hcp 1000000 4 2>/dev/null | cut -c1-420
  AMX_GCERT2_THIRD=$t python tools/r05/proto_fit.py 150 1000000 4 2>/dev/null | head -1
  AMX_GCERT2_THIRD=$t python tools/r05/proto_fit.py bench 1000000 6 2>/dev/null | cut -c1-420
  AMX_GCERT2_THIRD=$t python tools/r05/proto_fit.py bench 200000 6 2>/dev/null | head -1
done 2>&1 | tee $O/third_pass_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "protocol_shapes or kkt or parity" > $O/gpu_tests_t3.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_t3.txt
AMX_GCERT2_THIRD=1 timeout 900 python -m pytest tests -m gpu -x -q -k "kkt or parity" > $O/gpu_tests_t3b.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_t3b.txt
