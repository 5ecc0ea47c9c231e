#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt
bash tools/r04/ab.sh "50000 200000 500000 1000000" default nowarm 2>&1 | tee $O/ab_warm.txt
for t in 0 100000000; do
  export AMX_SEED_OCC2_FROM=$t
  echo "AMX_SEED_OCC2_FROM=$t"
  bash tools/r04/ab.sh "300000 500000 700000" default 2>&1 | tee -a $O/ab_occ_threshold.txt
done
unset AMX_SEED_OCC2_FROM
