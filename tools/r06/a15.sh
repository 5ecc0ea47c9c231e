#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a15; mkdir -p $O
for p in bench 105 150 hcp; do timeout -s KILL 300 python tools/r05/proto_fit.py $p 1000000 6 2>&1 | tail -2 | cut -c1-330; done | tee $O/protocols_final.txt
