#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a26; mkdir -p $O
timeout -s KILL 2000 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1
grep -n "passed\|failed\|Error\|error" $O/tests_all.txt | head -20
tail -40 $O/tests_all.txt | cut -c1-300
