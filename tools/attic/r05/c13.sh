#!/bin/bash
# third session, call 13: narrowing threads on stripes of cores (default) / one core each (1) / the node as a set (0), with the calling
# process confined to the device's node (taskset 0-63), to the other node (64-127), or free
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -2
for rep in 1 2; do for where in "0-63,128-191" "64-127,192-255" "0-255"; do for cores in 2 1 0; do
  echo "taskset $where cores $cores"
  AMX_HOST_PIN_CORES=$cores taskset -c $where timeout 300 python tools/r05/host_trace.py 1000000 6 2>&1 | grep "^float64 h"
done; done; done | tee $O/c13_stripes.txt
