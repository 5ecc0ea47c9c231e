#!/bin/bash
# protocols off the tuned point, before (variants/tabrm) and after the node consolidation
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a29; mkdir -p $O
for rep in 1 2; do
for lib in variants/tabrm/libamico_amd.so ""; do
  for p in 105 150 288; do
    echo -n "rep $rep lib '$lib' " | tee -a $O/ab.txt
    AMICO_AMD_LIB=$lib timeout -s KILL 300 python tools/r05/proto_fit.py $p 1000000 8 2>&1 | grep "volumes" | head -1 | cut -c1-220 | tee -a $O/ab.txt
  done
done
done
