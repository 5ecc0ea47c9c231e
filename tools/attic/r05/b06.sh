#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
bash tools/r04/ab.sh "50000 200000 1000000" default default 2>&1 | tee $O/ab_$1.txt
timeout 1700 python -m pytest tests -m gpu -x -q > $O/gpu_tests_$1.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_$1.txt; grep -E "^FAILED|Error" $O/gpu_tests_$1.txt | head
