#!/bin/bash
# where a certified left-over voxel's time goes in the wavefront-per-voxel kernels (variants/phases: -DAMX_PHASES)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a38; mkdir -p $O
AMICO_AMD_LIB=variants/phases/libamico_amd.so AB_STEPS=2 AB_PROFILING=0 timeout -s KILL 300 python tools/r06/fork_ab.py "1000000" "AMX_FORK=0" > $O/phases.txt 2>&1
grep -n "phases\|voxels:" $O/phases.txt | tail -8 | cut -c1-400
