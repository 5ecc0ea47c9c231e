#!/bin/bash
# fourth session, call 3: stress parity of the final sources on seeds the earlier collections did not use (HIP path vs the CPU oracle,
# 300 000 voxels per model and noise level; the hard signal mix)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
for seed in 2026 31337; do
  echo "# stress_parity.py 300000 $seed"
  timeout 900 python tools/stress_parity.py 300000 $seed 2>&1 | grep -v amdgpu.ids
  echo "# stress_hard.py 200000 $seed"
  timeout 900 python tools/stress_hard.py 200000 $seed 2>&1 | grep -v amdgpu.ids
done | tee $O/stress_parity.txt
