#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a27; mkdir -p $O
for p in 1 0 1 0; do
  echo "== profiling events $p" | tee -a $O/ab.txt
  AB_PROFILING=$p AB_STEPS=20 timeout -s KILL 300 python tools/r06/fork_ab.py "50000 200000 1000000" "AMX_FORK=0" 2>&1 | grep voxels | tee -a $O/ab.txt
done
for lib in variants/tabrm/libamico_amd.so ""; do
  echo "== freewater lib '$lib'" | tee -a $O/fw.txt
  AMICO_AMD_LIB=$lib timeout -s KILL 300 python bench.py --model freewater --voxels 2000000 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300 | tee -a $O/fw.txt
  AMICO_AMD_LIB=$lib timeout -s KILL 300 python bench.py --model freewater --voxels 200000 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300 | tee -a $O/fw.txt
done
