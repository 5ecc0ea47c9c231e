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
timeout 600 python -m pytest tests/test_signal.py -m gpu -x -q 2>&1 | tail -2
for n in 45000 50000 70000; do for f in 0 1000000000; do
AMX_SEED_OCC2_FROM=$f AMX_SEED2_OCC2_FROM=$f python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('occ2_from %-10s %8d voxels: %7.2f M voxels/s %7.3f ms | seed1 %.3f lasso_seed %.3f' % ('$f', $n, d['value']/1e6, d['ms_per_step'], r['seed_solver_ms'][0], r['seed_solver_ms'][1]))"
done; done
for n in 2000000 4000000 8000000; do for f in 0 1000000000 0 1000000000; do
AMX_RESCUE_FROM=$f python bench.py --steps 5 --warmup 2 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; sc=d.get('seed_chain', {})
print('rescue_from %-10s %8d voxels: %7.2f M voxels/s %7.3f ms groups s1 %.3f s2 %.3f s3 %.3f left %.3f %.3f %.3f | left %s' % ('$f', $n, d['value']/1e6, d['ms_per_step'], r['seed_ms'][0], r['seed_ms'][1], r['seed_ms'][2], r['stage_ms'][0], r['stage_ms'][1], r['stage_ms'][2], [sc.get(q) for q in ('leftover_stage1', 'leftover_lasso', 'leftover_stage3')]))"
done; done
