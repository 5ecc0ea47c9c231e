#!/bin/bash
# round 5, fourth GPU call: the 288-volume fit with batched global-tile sweeps + rescue pass; shape tests again
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solvers.py -m gpu -x -q -k "other_protocol_shapes or batched" 2>&1 | tail -3
timeout 600 rocprofv3 --kernel-trace --stats -d $O/hcp -o hcp -- python tools/r05/proto_fit.py hcp 1000000 3 > $O/hcp.log 2>&1; grep "^hcp\|^{" $O/hcp.log
python tools/rocpd_summary.py $O/hcp/hcp_results.db > $O/hcp_kernels.txt 2>&1; head -16 $O/hcp_kernels.txt | cut -c1-150
python tools/r05/proto_fit.py bench 1000000 5 2>&1 | grep "^bench\|^{"
