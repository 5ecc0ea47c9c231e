#!/bin/bash
# flakiness check of the final build: the GPU suite three times over (random order of files reversed on the second pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a36; mkdir -p $O
for i in 1 2 3; do
  timeout -s KILL 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests_$i.txt 2>&1; grep -n "passed\|failed" $O/tests_$i.txt | tail -2
done
