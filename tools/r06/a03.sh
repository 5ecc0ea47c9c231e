#!/bin/bash
# round 6, third GPU call: new tests (multi-device, sibling pools, fork, 181 volumes), calibration with the lone-line probes, default bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a03
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_boundary.py -m gpu -x -q > $O/gpu_tests_multi.txt 2>&1; tail -15 $O/gpu_tests_multi.txt
bash tools/r06/calib.sh > gpurun_out/r06_calib.log 2>&1; grep -A8 "k_read_line8\|k_read_row512" gpurun_out/r06_calib/summary.txt | grep -v "WRREQ\|WRITE\|TCP_" | head -40
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; tail -c 2200 $O/bench_line.json; tail -3 $O/bench_err.txt
