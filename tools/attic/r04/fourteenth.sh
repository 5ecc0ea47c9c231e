#!/bin/bash
# block-level work sharing in the table kernel and the certificates (BlockFeed) against the build before it
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests -m gpu -x -q -k "noddi or kkt or parity or fullsize" > gpurun_out/r04d/tests.txt 2>&1; tail -3 gpurun_out/r04d/tests.txt
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default head default head 2>&1 | tee gpurun_out/r04d/ab.txt
for v in default head; do
  unset AMICO_AMD_LIB; [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  echo "== $v"; python tools/r04/skew_ab.py 1000000 2>&1 | grep "voxels \|populations"
done 2>&1 | tee gpurun_out/r04d/skew.txt
