#!/bin/bash
# third session, call 9: the whole-call narrowing job (ring of four chunks): boundary tests, then host_trace.py and bench.py under the three placements
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -3 | tee $O/c09_tests.txt
for pin in caller gpu 0; do
  echo "pin $pin"
  AMX_HOST_PIN=$pin timeout 300 python tools/r05/host_trace.py 1000000 8 2>&1 | grep "^float64 h"
  AMX_HOST_PIN=$pin AMX_HOST_TRACE=1 timeout 900 python bench.py --no-cpu-baseline > $O/c09_bench_$pin.json 2> $O/c09_trace_$pin.txt
  python - <<PY
import json
d=json.loads(open('$O/c09_bench_$pin.json').read().strip().splitlines()[-1]); o=d['other_configs']
print('bench.py', {k: (round(o[k]['ms_per_call'],2), o[k].get('batches_as_float32')) for k in o if k.startswith('noddi_host')}, {k: {kk: round(vv['value']/1e6,1) for kk, vv in o[k]['host_buffers'].items()} for k in ('freewater_2M', 'sandi_1M') if isinstance(o[k].get('host_buffers'), dict)})
PY
  grep "amx host trace" $O/c09_trace_$pin.txt | sed -n 11,15p
done | tee $O/c09_summary.txt
