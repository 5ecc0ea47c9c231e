#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a28; mkdir -p $O
timeout -s KILL 1500 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt; tail -c 2500 $O/bench_line.json; echo; tail -5 $O/bench_err.txt
