#!/bin/bash
# third session, call 6: kernel traces of mid-size calls (300 000 and 100 000 voxels) next to the 1 M trace: which kernels hold the floor
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r05/ktrace.sh r05c_300k --voxels 300000 > /dev/null
bash tools/r05/ktrace.sh r05c_100k --voxels 100000 > /dev/null
cut -c1-150 gpurun_out/r05c_300k_kernel_stats.txt | head -22
cut -c1-150 gpurun_out/r05c_100k_kernel_stats.txt | head -22
