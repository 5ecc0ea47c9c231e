#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a31; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_czb.py -x -q -m gpu > $O/tests.txt 2>&1; grep -n "passed\|failed" $O/tests.txt; tail -30 $O/tests.txt | cut -c1-200
