#!/bin/bash
# A/B of library variants at several call sizes: bash tools/r04/ab.sh "<sizes>" default v1 v2 ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
sizes=$1; shift
for v in "$@"; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  for n in $sizes; do
    python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-8s %8d voxels: %7.2f M voxels/s %7.3f ms | seed1 %.3f lasso_seed %.3f | groups s1 %.3f s2 %.3f s3 %.3f left %.3f %.3f %.3f | dmap %.1e rerun %d' % ('$v', $n, d['value']/1e6, d['ms_per_step'], r['seed_solver_ms'][0], r['seed_solver_ms'][1], r['seed_ms'][0], r['seed_ms'][1], r['seed_ms'][2], r['stage_ms'][0], r['stage_ms'][1], r['stage_ms'][2], d['parity']['max_abs_dmap'], d['solver_stats']['rerun_voxels']))"
  done
done
unset AMICO_AMD_LIB
