#!/bin/bash
# does the last partial round of workgroups show in the NODDI stage times?  voxel counts around a multiple of 256 chunks
for n in 950000 975000 985000 1000000 1015000 1030000 1048000; do
  python bench.py --voxels $n --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['roofline']['stage_ms']
        print('$n', 'ns/voxel %.3f' % (1e6 * d['ms_per_step'] / $n), 'stage ms', ['%.2f' % v for v in s], 'ns/voxel per stage', ['%.3f' % (1e6 * v / $n) for v in s])
"
done
