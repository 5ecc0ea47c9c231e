#!/bin/bash
# first certificate pass of every stage beside its seed solver (second stream, chunk-level hand-over) against AMX_NO_OVERLAP=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
for rep in 1 2; do
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default 2>&1
AMX_NO_OVERLAP=1 bash tools/r04/ab.sh "50000 200000 1000000 4000000" default 2>&1 | sed 's/^default/serial /'
done | tee gpurun_out/r04f/ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "noddi or kkt or parity or skewed or repeatable or fullsize" > gpurun_out/r04f/tests.txt 2>&1; tail -4 gpurun_out/r04f/tests.txt
python tools/r04/skew_ab.py 1000000 2>&1 | grep "voxels "
