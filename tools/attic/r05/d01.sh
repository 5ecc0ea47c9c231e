#!/bin/bash
# fourth session, call 1: default bench line of the final sources (the committed r05c line was of the build before the stripes) and the
# host-buffer call in eight fresh processes (does the 28 ms case still occur?)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; tail -c 300 $O/bench_line.json
for rep in 1 2 3 4 5 6 7 8; do
  timeout 300 python tools/r05/host_trace.py 1000000 6 2>&1 | grep "^float"
done | tee $O/d01_host_processes.txt
