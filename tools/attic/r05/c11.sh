#!/bin/bash
# third session, call 11: the knobs of the seeded chain at the sizes the host batches run at (300 000 / 150 000 voxels)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
run() { env "$@" python bench.py --steps 10 --warmup 3 --voxels $V --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['seed_chain']; print('%-44s %d voxels: %.3f ms  left-overs %d / %d / %d  max|dmap| %.1e' % ('$*', d['config']['voxels_per_gpu'], d['ms_per_step'], s['leftover_stage1'], s['leftover_lasso'], s['leftover_stage3'], d['parity']['max_abs_dmap']))"; }
for V in 300000 150000; do
  run A=0
  run AMX_SEED_TRIPCAP=16,16,8
  run AMX_SEED_TRIPCAP=14,14,7
  run AMX_SEED_TRIPCAP=12,12,6
  run AMX_SEED_TRIPCAP=28,24,12
  run AMX_SEED_OCC2_FROM=1000000 AMX_SEED2_OCC2_FROM=1000000
  run AMX_SEED_CHUNK=512
  run AMX_SEED_CHUNK=1024
  run AMX_SEED_WAVES=2
  run A=0
done | tee $O/c11_knobs.txt
