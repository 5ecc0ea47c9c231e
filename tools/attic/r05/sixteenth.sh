#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05p
timeout 900 python tools/stress_parity.py 300000 123 > gpurun_out/r05p/stress_parity.txt 2>&1; tail -12 gpurun_out/r05p/stress_parity.txt
timeout 900 python tools/stress_hard.py 300000 123 > gpurun_out/r05p/stress_hard.txt 2>&1; tail -6 gpurun_out/r05p/stress_hard.txt
