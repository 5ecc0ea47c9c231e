#!/bin/bash
# where the seeded chain starts to pay under the shorter floor (AMX_SEED_MIN_VOXELS: 0 = always, 10^9 = never)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 10000 15000 20000 25000 30000 40000; do for f in 0 1000000000; do
AMX_SEED_MIN_VOXELS=$f python bench.py --steps 10 --warmup 3 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('seed_min_voxels %-10s %8d voxels: %7.2f M voxels/s %7.3f ms dmap %.1e' % ('$f', $n, d['value']/1e6, d['ms_per_step'], d['parity']['max_abs_dmap']))"
done; done
