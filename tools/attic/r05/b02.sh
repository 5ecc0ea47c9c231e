#!/bin/bash
# round 5, second session: the LASSO certificates' Gram / table gathers as unconditional loads (all in flight) against the build of c696c5c
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
bash tools/r04/ab.sh "50000 200000 1000000" base default base default 2>&1 | tee $O/ab_gcert2.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or kkt or multi" > $O/gpu_tests_part2.txt 2>&1; tail -3 $O/gpu_tests_part2.txt
