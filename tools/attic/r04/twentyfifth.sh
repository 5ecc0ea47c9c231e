#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 2000000 8000000; do
python bench.py --steps 5 --warmup 2 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print($n, '%.2f M voxels/s' % (d['value'] / 1e6), d.get('seed_chain'))"
AMX_RESCUE_FROM=100000000 python bench.py --steps 5 --warmup 2 --voxels $n --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print($n, 'no rescue %.2f M voxels/s' % (d['value'] / 1e6), d.get('seed_chain'))"
done
