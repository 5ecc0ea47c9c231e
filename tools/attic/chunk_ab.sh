#!/bin/bash
for v in "" ch192 ch384 ch512; do
  if [ -n "$v" ]; then export AMICO_AMD_LIB=variants/$v/libamico_amd.so; else unset AMICO_AMD_LIB; fi
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('${v:-default}', '%.2f M voxels/s' % (d['value'] / 1e6), d['roofline']['stage_ms'])
"
done
