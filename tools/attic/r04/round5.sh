#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in prev default prev default; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  timeout 300 python bench.py --model prep --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); L=d['layouts']
print('$v', ' | '.join('%s %.3f ms %.0f GB/s (f32 rows %.3f ms) exact %s' % (k, L[k]['kernel_ms'], L[k]['achieved_GBs'], L[k].get('float32_rows', {}).get('kernel_ms', 0), L[k]['bit_exact_vs_numpy']) for k in L))"
done
unset AMICO_AMD_LIB
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/round5_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/round5_tests.txt
timeout 900 python tools/r04/host_sweep.py 1000000 2>&1 | grep batch
