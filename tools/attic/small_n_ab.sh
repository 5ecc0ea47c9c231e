#!/bin/bash
# NODDI throughput at clinical voxel counts, chunk sizes 256 (default) / 128 / 64
for n in 50000 100000 200000 400000; do
for v in "" ch128 ch64; do
  if [ -n "$v" ]; then export AMICO_AMD_LIB=variants/$v/libamico_amd.so; else unset AMICO_AMD_LIB; fi
  python bench.py --voxels $n --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$n', '${v:-ch256}', '%.2f M voxels/s' % (d['value'] / 1e6), ['%.2f' % x for x in d['roofline']['stage_ms']])
"
done; done
