#!/bin/bash
# tile columns with their loads in flight: global tiles (288 volumes) in the default build, LDS tiles too in variants/collds
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
for v in base default collds; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  echo "== $v"
  python tools/r05/proto_fit.py hcp 1000000 4 2>/dev/null | head -1
  python tools/r05/proto_fit.py bench 1000000 6 2>/dev/null | head -1
  python tools/r05/proto_fit.py bench 200000 6 2>/dev/null | head -1
done 2>&1 | tee $O/tile_column_ab.txt
unset AMICO_AMD_LIB
timeout 900 python -m pytest tests -m gpu -x -q -k "protocol_shapes or solvers or kkt" > $O/gpu_tests_tc.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_tc.txt
