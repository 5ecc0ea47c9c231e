#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py tests/test_gpu_boundary.py tests/test_gpu_solvers.py -m gpu -x -q -k "noddi or solver or batched or lasso or nnls or float32 or lambda" > $O/tests.txt 2>&1; grep -n "passed\|failed\|Error\|assert" $O/tests.txt | head
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default 2>&1 | tee $O/ab.txt
AMX_NO_GCERT_WIDE=1 bash tools/r04/ab.sh "1000000" default 2>&1 | tee -a $O/ab.txt
