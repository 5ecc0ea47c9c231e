#!/bin/bash
# seed chunk size / wavefronts per workgroup once more, under the trip caps and the work sharing
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 200000 1000000 4000000; do for c in 0 1024 2048 8192 16384; do for w in 0 2; do
AMX_SEED_CHUNK=$c AMX_SEED_WAVES=$w python bench.py --steps 6 --warmup 2 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('chunk %-6s waves %s %8d voxels: %7.2f M voxels/s %7.3f ms | seed1 %.3f lasso_seed %.3f groups s1 %.3f s2 %.3f s3 %.3f left %.3f %.3f %.3f' % ('$c', '$w', $n, d['value']/1e6, d['ms_per_step'], r['seed_solver_ms'][0], r['seed_solver_ms'][1], r['seed_ms'][0], r['seed_ms'][1], r['seed_ms'][2], r['stage_ms'][0], r['stage_ms'][1], r['stage_ms'][2]))"
done; done; done
