#!/bin/bash
# cap on the flagged atoms a lane of k_nnls_gcert examines exactly (AMX_GCERT_MAX_FLAG); above it the voxel goes to the left-over kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 200000 1000000; do for c in 1000000 48 24 12 6 1000000; do
AMX_GCERT_MAX_FLAG=$c python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; sc=d.get('seed_chain', {})
print('max_flag %-8s %8d voxels: %7.2f M voxels/s %7.3f ms groups s1 %.3f s2 %.3f s3 %.3f left %.3f %.3f %.3f | left %s dmap %.1e' % ('$c', $n, d['value']/1e6, d['ms_per_step'], r['seed_ms'][0], r['seed_ms'][1], r['seed_ms'][2], r['stage_ms'][0], r['stage_ms'][1], r['stage_ms'][2], [sc.get(q) for q in ('leftover_stage1', 'leftover_lasso', 'leftover_stage3')], d['parity']['max_abs_dmap']))"
done; done
