#!/bin/bash
# fewer nodes per fit (counters cleared in one launch, chunk order in k_plan's tail, no output memset, one-kernel status sync, k_dir_to_lut span by call size)
# against the library before (variants/tabrm = the sources of commit 33b9bd3): tests, fit times by size, timeline of a 100 k fit
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a25; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "writes_every_voxel or errors_and_edge or exvivo" 2>&1 | tail -5 | tee $O/tests_first.txt
for rep in 1 2; do
for lib in variants/tabrm/libamico_amd.so ""; do
  echo "== rep $rep lib '$lib'" | tee -a $O/ab.txt
  AMICO_AMD_LIB=$lib AB_STEPS=20 timeout -s KILL 300 python tools/r06/fork_ab.py "50000 100000 200000 300000 1000000" "AMX_FORK=0" 2>&1 | grep voxels | tee -a $O/ab.txt
done
done
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $O/tests_all.txt
