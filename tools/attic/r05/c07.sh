#!/bin/bash
# default bench line (counter figures of the same build) + stress parity against the oracle
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py > gpurun_out/bench_line_r05c.json 2> gpurun_out/bench_err_r05c.txt; tail -c 200 gpurun_out/bench_line_r05c.json
timeout 1500 python tools/stress_parity.py > gpurun_out/stress_parity_r05c.txt 2>&1; tail -12 gpurun_out/stress_parity_r05c.txt
