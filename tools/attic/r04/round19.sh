#!/bin/bash
# k_prep_gather's direct path with half tiles (default) against full tiles (AMX_PREP_FULL_TILES=1) and the tree before (variants/prev)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --model prep --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); L=d['layouts']
print('$1', ' | '.join('%s %.3f ms %.0f GB/s (f32 rows %.3f ms) exact %s' % (k, L[k]['kernel_ms'], L[k]['achieved_GBs'], L[k].get('float32_rows', {}).get('kernel_ms', 0), L[k]['bit_exact_vs_numpy']) for k in L))"; }
for rep in 1 2; do
  AMICO_AMD_LIB=$PWD/variants/prev/libamico_amd.so run prev
  run half
  AMX_PREP_FULL_TILES=1 run full
done
timeout 600 python -m pytest tests/test_signal.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep "passed\|failed"
