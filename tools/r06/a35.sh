#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a35; mkdir -p $O
AMX_DEBUG=1 AB_STEPS=1 AB_PROFILING=0 timeout -s KILL 300 python tools/r06/fork_ab.py "1000000" "AMX_FORK=0" > $O/dbg.txt 2>&1
grep -n "dual-vector\|voxels:" $O/dbg.txt | tail -6 | cut -c1-300
