#!/bin/bash
# third session, call 10: one narrowing thread per physical core (AMX_HOST_PIN_CORES) against the node's CPUs as a set; host_trace.py and bench.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
for rep in 1 2; do for cfg in "caller 1" "gpu 1" "caller 0"; do
  set -- $cfg
  echo "pin $1 cores $2"
  AMX_HOST_PIN=$1 AMX_HOST_PIN_CORES=$2 timeout 300 python tools/r05/host_trace.py 1000000 8 2>&1 | grep "^float64 h"
  AMX_HOST_PIN=$1 AMX_HOST_PIN_CORES=$2 timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['other_configs']
print('bench.py', {k: (round(o[k]['ms_per_call'],2), o[k].get('batches_as_float32')) for k in o if k.startswith('noddi_host')}, {k: {kk: round(vv['value']/1e6,1) for kk, vv in o[k]['host_buffers'].items()} for k in ('freewater_2M', 'sandi_1M') if isinstance(o[k].get('host_buffers'), dict)})"
done; done | tee $O/c10_summary.txt
