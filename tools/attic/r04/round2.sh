#!/bin/bash
# (1) signal preparation with live-tile lists against the previous build; (2) the seeded chain with the new trip caps: all bench legs against
# the caps of 64; (3) occupancy / rescue thresholds under the new caps
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
for c in 64,64,64 28,24,12; do
AMX_SEED_TRIPCAP=$c timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); o = d['other_configs']
print('caps $c headline %.1f' % (d['value'] / 1e6), ' '.join('%s %.1f (left %s)' % (k, o[k]['value'] / 1e6, [o[k]['seed_chain'][q] for q in ('leftover_stage1', 'leftover_lasso', 'leftover_stage3')]) for k in ('noddi_hard_mix', 'noddi_105vol', 'noddi_150vol', 'noddi_exvivo')), ' host %.1f / %.1f' % (o['noddi_host_buffers']['value'] / 1e6, o['noddi_host_buffers_f32']['value'] / 1e6))"
done
for n in 100000 150000 200000 300000; do for f in 0 1000000000; do
AMX_SEED_OCC2_FROM=$f AMX_SEED2_OCC2_FROM=$f python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('occ2_from %-10s %8d voxels: %7.2f M voxels/s %7.3f ms | seed1 %.3f lasso_seed %.3f' % ('$f', $n, d['value']/1e6, d['ms_per_step'], r['seed_solver_ms'][0], r['seed_solver_ms'][1]))"
done; done
for n in 500000 1000000 2000000; do for f in 0 1000000000; do
AMX_RESCUE_FROM=$f python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('rescue_from %-10s %8d voxels: %7.2f M voxels/s %7.3f ms' % ('$f', $n, d['value']/1e6, d['ms_per_step']))"
done; done
