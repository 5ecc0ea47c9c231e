#!/bin/bash
# the rescue pass of the NNLS certificates at every size, after its staging got loads in flight
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
for rf in 2000000 0; do
  for n in 200000 1000000 2000000; do
    AMX_RESCUE_FROM=$rf python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('rescue_from %8d %8d voxels: %7.2f M voxels/s %7.3f ms | groups s1 %.3f s2 %.3f s3 %.3f left %.3f %.3f %.3f | dmap %.1e' % ($rf, $n, d['value']/1e6, d['ms_per_step'], r['seed_ms'][0], r['seed_ms'][1], r['seed_ms'][2], r['stage_ms'][0], r['stage_ms'][1], r['stage_ms'][2], d['parity']['max_abs_dmap']))"
  done
done 2>&1 | tee $O/rescue_ab.txt
