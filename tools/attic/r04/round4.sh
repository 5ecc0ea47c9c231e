#!/bin/bash
# prep (per-tile tickets requested ahead, 64 loads in flight) against the previous build; hard-first left-overs A/B; FreeWater static first unit; KKT + parity tests
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in prev default prev default; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  timeout 300 python bench.py --model prep --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); L=d['layouts']
print('$v', ' | '.join('%s %.3f ms %.0f GB/s (f32 rows %.3f ms) exact %s' % (k, L[k]['kernel_ms'], L[k]['achieved_GBs'], L[k].get('float32_rows', {}).get('kernel_ms', 0), L[k]['bit_exact_vs_numpy']) for k in L))"
  timeout 300 python bench.py --model freewater --voxels 2000000 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v freewater 2M: %.1f M voxels/s %.3f ms kernel_ms %s' % (d['value']/1e6, d['ms_per_step'], d.get('kernel_ms')))"
done
unset AMICO_AMD_LIB
timeout 600 python -m pytest tests/test_signal.py -m gpu -x -q 2>&1 | tail -2
for n in 50000 200000 1000000 4000000; do for f in 1 0 1 0; do
AMX_NO_HARD_FIRST=$f python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; sc=d.get('seed_chain', {})
print('no_hard_first=$f %8d voxels: %7.2f M voxels/s %7.3f ms groups s1 %.3f s2 %.3f s3 %.3f left %.3f %.3f %.3f | left %s dmap %.1e' % ($n, d['value']/1e6, d['ms_per_step'], r['seed_ms'][0], r['seed_ms'][1], r['seed_ms'][2], r['stage_ms'][0], r['stage_ms'][1], r['stage_ms'][2], [sc.get(q) for q in ('leftover_stage1', 'leftover_lasso', 'leftover_stage3')], d['parity']['max_abs_dmap']))"
done; done
timeout 1500 python -m pytest tests -m gpu -x -q -k "kkt or parity or multi" 2>&1 | tail -3
