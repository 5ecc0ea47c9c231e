#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a33; mkdir -p $O
AMX_HOST_TRACE=1 timeout -s KILL 300 python tools/r05/host_trace.py 1000000 4 > $O/host_trace.txt 2>&1
grep -v "^---" $O/host_trace.txt | tail -80 | cut -c1-200
