#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_czb.py -m gpu -x -q > $O/new_tests.txt 2>&1; grep -n "passed\|failed\|Error\|assert" $O/new_tests.txt | head -20
bash tools/r04/ab.sh "200000 1000000" default 2>&1 | tee $O/ab.txt
AMX_SEED_OCC2_FROM=0 AMX_SEED_WAVES=4 bash tools/r04/ab.sh "100000 200000 300000" default 2>&1 | tee $O/ab_occ2_4waves.txt
