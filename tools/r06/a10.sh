#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a10
mkdir -p $O
for p in bench 105 150 hcp; do
  for sw in "X=0"; do
    echo "== $p, $sw"
    env $sw timeout -s KILL 300 python tools/r05/proto_fit.py $p 1000000 5 2>&1 | grep -v "^$" | tail -2 | cut -c1-330
  done
done | tee $O/protocols_after.txt
for n in 300000 100000; do echo "== 105, $n voxels"; timeout -s KILL 200 python tools/r05/proto_fit.py 105 $n 5 2>&1 | tail -2 | head -1 | cut -c1-250; AMX_GCERT2_THIRD=0 timeout -s KILL 200 python tools/r05/proto_fit.py 105 $n 5 2>&1 | tail -2 | head -1 | cut -c1-250; done | tee -a $O/protocols_after.txt
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kkt.py -m gpu -q -x 2>&1 | tail -3
