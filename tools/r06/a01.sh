#!/bin/bash
# round 6, first GPU call: the tree with the ADVICE fixes (suite), then the left-over fork A/B (AMX_FORK) against the unforked chain
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a01
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout 900 python tools/r06/fork_ab.py "50000 100000 200000 300000 1000000" "AMX_FORK=0" "AMX_FORK=2" "AMX_FORK=2 AMX_FORK_PRIO=1" "AMX_FORK=1" "AMX_FORK=3" "AMX_FORK=3 AMX_FORK_PRIO=1" "AMX_FORK=3 AMX_FORK_CUS=64" > $O/fork_ab.txt 2>&1
cat $O/fork_ab.txt | tail -40
AMICO_AMD_LIB=$PWD/variants/r05/libamico_amd.so timeout 300 python tools/r06/fork_ab.py "50000 100000 200000 300000 1000000" "AMX_FORK=0" > $O/r05_lib.txt 2>&1; tail -5 $O/r05_lib.txt
