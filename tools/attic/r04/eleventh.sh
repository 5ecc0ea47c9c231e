#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_parity.py -m gpu -x -q -s -k "exvivo or hard" > $O/tests.txt 2>&1; grep -n "passed\|failed\|Error\|assert\|hard mix\|max " $O/tests.txt | head -20
