#!/bin/bash
# more stress parity of the final build: two more seeds for the four models and the hard mix, one more for the six protocols
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a41; mkdir -p $O
{
python -c "from amico_amd import _capi; print('#', _capi.build_id())"
for seed in 31337 2718; do
  echo "# stress_parity.py 300000 $seed"; timeout -s KILL 900 python tools/stress_parity.py 300000 $seed 2>&1 | grep -v "amdgpu.ids"
  echo "# stress_hard.py 200000 $seed"; timeout -s KILL 900 python tools/stress_hard.py 200000 $seed 2>&1 | grep -v "amdgpu.ids"
done
echo "# stress_protocols.py 1000000 5 23"
timeout -s KILL 1500 python tools/r06/stress_protocols.py 1000000 5 23 2>&1 | grep -v amdgpu.ids | cut -c1-400
} | tee $O/stress_more.txt | cut -c1-170
