#!/bin/bash
# a second look at mendable seeds in the NNLS certificates (k_nnls_gcert<.., REPAIR>): 288 volumes and the bench shape
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
for r in 0 1; do
  echo "== AMX_GCERT_REPAIR=$r"
  AMX_GCERT_REPAIR=$r python tools/r05/proto_fit.py hcp 1000000 4 2>/dev/null | cut -c1-330
  AMX_GCERT_REPAIR=$r python tools/r05/proto_fit.py 150 1000000 4 2>/dev/null | cut -c1-330
  AMX_GCERT_REPAIR=$r python tools/r05/proto_fit.py bench 1000000 6 2>/dev/null | cut -c1-330
done 2>&1 | tee $O/repair_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "protocol_shapes" 2>&1 | tail -2
AMX_GCERT_REPAIR=1 timeout 1200 python -m pytest tests -m gpu -x -q -k "kkt or parity or fullsize" > $O/gpu_tests_rep.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_rep.txt; grep -E "^FAILED|^E  " $O/gpu_tests_rep.txt | head -5
