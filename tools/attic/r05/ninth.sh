#!/bin/bash
# round 5, ninth GPU call: LASSO certificates on the lean lane state (one triangle, in-place solve): wide pass at 18 / 16 / 16+18 / 17 atoms
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
bash tools/r04/ab.sh "200000 1000000" default w16 w16_18 w17 2>&1 | tee $O/ab.txt
for v in default w16; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/k_$v -o k -- python tools/r05/proto_fit.py bench 1000000 3 > $O/k_$v.log 2>&1
  grep "^{" $O/k_$v.log | cut -c1-200
  python tools/rocpd_summary.py $O/k_$v/k_results.db | grep "lasso_gcert\|k_noddi<4" | cut -c1-130
done
