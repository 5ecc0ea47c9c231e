#!/bin/bash
# round 5, first GPU call: the suite at the hygiene commit + what per-chunk isolation costs (no work sharing between chunks)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
bash tools/r04/ab.sh "50000 200000 1000000" default nosteal 2>&1 | tee $O/ab_nosteal.txt
