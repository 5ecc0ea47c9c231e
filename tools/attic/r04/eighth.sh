#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt
bash tools/r04/profile.sh r04a
bash tools/profile_small.sh r04a
