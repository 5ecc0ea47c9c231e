#!/bin/bash
# third session, call 8: the host-buffer legs INSIDE bench.py (17.4 ms there against 12.5 ms in tools/r05/host_trace.py): timeline, thread placement
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
for pin in caller 0 gpu; do
  AMX_HOST_PIN=$pin AMX_HOST_TRACE=1 timeout 900 python bench.py --no-cpu-baseline > $O/c08_bench_$pin.json 2> $O/c08_trace_$pin.txt
  echo "pin $pin"; python - <<PY
import json
d=json.loads(open('$O/c08_bench_$pin.json').read().strip().splitlines()[-1]); o=d['other_configs']
print({k: (round(o[k]['ms_per_call'],2), o[k].get('batches_as_float32')) for k in o if k.startswith('noddi_host')})
PY
  grep "amx host trace" $O/c08_trace_$pin.txt | sed -n 6,20p
done
