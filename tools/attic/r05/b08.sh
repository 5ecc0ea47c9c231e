#!/bin/bash
# trip caps of the seed solvers once more: the LASSO left-over solver now starts from the seed solver's passive set, stage 1 hands its support on
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
for caps in ${CAPS:-24,24,10 22,24,10 20,24,10 18,24,10 20,20,10 20,24,8}; do
  for n in 50000 200000 1000000; do
    AMX_SEED_TRIPCAP=$caps python bench.py --steps 8 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('caps %-9s %8d voxels: %7.2f M voxels/s %7.3f ms | seed1 %.3f lasso_seed %.3f | groups s1 %.3f s2 %.3f s3 %.3f left %.3f %.3f %.3f | dmap %.1e' % ('$caps', $n, d['value']/1e6, d['ms_per_step'], r['seed_solver_ms'][0], r['seed_solver_ms'][1], r['seed_ms'][0], r['seed_ms'][1], r['seed_ms'][2], r['stage_ms'][0], r['stage_ms'][1], r['stage_ms'][2], d['parity']['max_abs_dmap']))"
  done
done 2>&1 | tee -a $O/tripcaps2.txt
