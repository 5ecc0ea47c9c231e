#!/bin/bash
# trip caps once more under the new entering rule / iso-first (fewer trips per voxel)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 50000 200000 1000000; do
for c in 28,24,12 24,24,12 20,24,12 24,24,10 22,20,10 28,24,12; do
AMX_SEED_TRIPCAP=$c python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; sc=d.get('seed_chain', {})
print('cap %-9s %8d voxels: %7.2f M voxels/s %7.3f ms | seed1 %.3f lasso_seed %.3f | groups s1 %.3f s2 %.3f s3 %.3f left %.3f %.3f %.3f | left %s' % ('$c', $n, d['value']/1e6, d['ms_per_step'], r['seed_solver_ms'][0], r['seed_solver_ms'][1], r['seed_ms'][0], r['seed_ms'][1], r['seed_ms'][2], r['stage_ms'][0], r['stage_ms'][1], r['stage_ms'][2], [sc.get(q) for q in ('leftover_stage1', 'leftover_lasso', 'leftover_stage3')]))"
done
done
