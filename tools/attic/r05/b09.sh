#!/bin/bash
# the second LASSO certificate pass with its lists shared out (BlockFeed) against own lists only
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
for ns in 1 0 1 0; do
  for n in 200000 1000000 4000000; do
    AMX_NO_WIDE_SHARE=$ns python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('no_wide_share=$ns %8d voxels: %7.2f M voxels/s %7.3f ms | groups s1 %.3f s2 %.3f s3 %.3f left %.3f %.3f %.3f | dmap %.1e' % ($n, d['value']/1e6, d['ms_per_step'], r['seed_ms'][0], r['seed_ms'][1], r['seed_ms'][2], r['stage_ms'][0], r['stage_ms'][1], r['stage_ms'][2], d['parity']['max_abs_dmap']))"
  done
done 2>&1 | tee $O/wide_share_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or kkt or multi" > $O/gpu_tests_ws.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_ws.txt
