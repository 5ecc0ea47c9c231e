#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a24; mkdir -p $O
timeout -s KILL 1500 python tools/r06/stress_protocols.py 1000000 5 11 2>&1 | grep -v amdgpu.ids | tee $O/stress_protocols.txt
