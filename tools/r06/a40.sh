#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a40; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "no_valid_direction" > $O/tests.txt 2>&1; tail -15 $O/tests.txt | cut -c1-200
