#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a08
mkdir -p $O
timeout -s KILL 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt | head -3
timeout -s KILL 200 python tools/r06/fork_ab.py "50000 100000 200000 300000 1000000" "AMX_FORK=0" 2>&1 | grep "^AMX\|fault"
