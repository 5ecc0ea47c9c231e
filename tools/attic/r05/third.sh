#!/bin/bash
# round 5, third GPU call: where the 288-volume fit spends its time (kernel trace), the fixed table GEMM (no spills) at the headline
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_solvers.py -m gpu -x -q 2>&1 | tail -3
bash tools/r04/ab.sh "200000 1000000" default 2>&1 | tee $O/ab.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/hcp -o hcp -- python tools/r05/proto_fit.py hcp 1000000 3 > $O/hcp.log 2>&1; tail -3 $O/hcp.log
python tools/rocpd_summary.py $(ls $O/hcp/*/*.db | head -1) > $O/hcp_kernels.txt 2>&1; head -30 $O/hcp_kernels.txt | cut -c1-200
