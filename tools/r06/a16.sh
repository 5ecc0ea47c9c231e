#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_a16
for pf in 0 1; do AMX_HOST_PREFETCH=$pf AMX_HOST_TRACE=1 timeout -s KILL 200 python tools/r06/first_call_parts.py 2>&1 | grep "PARTS\|host trace" | head -40; done | tee gpurun_out/r06_a16/first_call_parts.txt
