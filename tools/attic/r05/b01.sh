#!/bin/bash
# round 5, second session, first GPU call: staging of the chunk tables with loads in flight, no serial look at y~ when a lane takes a voxel,
# the last batch of a float64 host call short -- against the build of c696c5c (variants/base)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
bash tools/r04/ab.sh "50000 200000 1000000" base default base default 2>&1 | tee $O/ab_stage.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "nan or finite or parity or kkt" > $O/gpu_tests_part.txt 2>&1; tail -3 $O/gpu_tests_part.txt
timeout 600 python tools/r05/host_tail.py 1000000 2>&1 | tee $O/host_tail.txt
