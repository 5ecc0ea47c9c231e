#!/bin/bash
# third session, call 12: from how many voxels a host-buffer call should travel in batches, now that the float64 copy is half as long
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
for n in 300000 400000 500000; do for from in 524288 262144; do
  echo "n $n pipeline from $from"; AMX_HOST_PIPELINE_FROM=$from timeout 300 python tools/r05/host_trace.py $n 8 2>&1 | grep "^float\|^genuine"
done; done | tee $O/c12_pipeline_from.txt
