#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a04
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_solvers.py -m gpu -x -q -s -k "dense or 64_atom" > $O/tests_big.txt 2>&1; tail -30 $O/tests_big.txt
