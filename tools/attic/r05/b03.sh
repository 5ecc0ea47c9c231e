#!/bin/bash
# A/B at three sizes against variants/base + kernel stats of the default build: bash tools/r05/b03.sh <tag> [pytest -k expression]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1
O=gpurun_out/r05b
mkdir -p $O
bash tools/r04/ab.sh "50000 200000 1000000" base default default 2>&1 | tee $O/ab_$TAG.txt
bash tools/r05/ktrace.sh $TAG > /dev/null 2>&1
grep -E "k_nnls|k_lasso|k_noddi" gpurun_out/${TAG}_kernel_stats.txt | cut -c1-125
if [ -n "$2" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$2" > $O/gpu_tests_$TAG.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_$TAG.txt; fi
