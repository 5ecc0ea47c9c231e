#!/bin/bash
# small-call builds of the left-over kernels (two workgroups per CU): each stage's build at every size against the round-5 builds
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a05
mkdir -p $O
B=2000000000
timeout 1200 python tools/r06/fork_ab.py "50000 100000 200000 300000 500000 1000000" "AMX_LEFT_SMALL=0,0,0" "AMX_LEFT_SMALL=$B,0,0" "AMX_LEFT_SMALL=0,$B,0" "AMX_LEFT_SMALL=0,0,$B" "AMX_LEFT_SMALL=$B,$B,$B" > $O/left_small_ab.txt 2>&1
grep "^AMX" $O/left_small_ab.txt
