#!/bin/bash
# round 5, twelfth GPU call: the whole suite on the tree with the fused FreeWater kernel and the 8-row FreeWater / CZB kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt; grep -n "Error\|error\|FAILED" $O/gpu_tests.txt | head
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
