#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_a18

for rep in 1 2 3; do
  for pf in 1 0; do AMX_HOST_PREFETCH=$pf timeout -s KILL 200 python tools/r05/model_fit_overhead.py 2>&1 | grep "voxels:" | sed "s/^/prefetch $pf: /"; done
done | tee gpurun_out/r06_a18/model_fit_overhead.txt
