#!/bin/bash
# third session, call 4: A/B in one box -- host threads that spin between jobs (AMX_HOST_SPIN_US) against threads that sleep
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
for rep in 1 2; do for spin in 0 1000 100; do
  echo "spin $spin us"; AMX_HOST_SPIN_US=$spin timeout 300 python tools/r05/host_trace.py 1000000 8 2>&1 | grep "^float64 h"
done; done | tee $O/c04_spin.txt
numactl --hardware 2>/dev/null | head -5; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3
