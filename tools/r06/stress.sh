#!/bin/bash
# stress parity of the final build on seeds no earlier collection used (the seed also draws the gradient scheme): bash tools/r06/stress.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_stress; mkdir -p $O
{
python -c "from amico_amd import _capi; print('#', _capi.build_id())"
for seed in 606 90210; do
  echo "# stress_parity.py 300000 $seed"; timeout -s KILL 900 python tools/stress_parity.py 300000 $seed 2>&1 | grep -v "amdgpu.ids"
  echo "# stress_hard.py 200000 $seed"; timeout -s KILL 900 python tools/stress_hard.py 200000 $seed 2>&1 | grep -v "amdgpu.ids"
done
echo "# the same NODDI voxels with the LASSO left-overs on a side stream (AMX_FORK=2)"
AMX_FORK=2 timeout -s KILL 900 python tools/stress_parity.py 300000 606 2>&1 | grep "^NODDI"
} | tee $O/stress_parity.txt
