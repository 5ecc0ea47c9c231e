#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for sw in "AMX_LEFT_NR4_NW8=0" "X=0"; do echo -n "rep $rep $sw: "; env $sw timeout -s KILL 300 python tools/r05/proto_fit.py 150 1000000 6 2>&1 | tail -2 | head -1 | cut -c1-230; done; done
for sw in "AMX_LEFT_NR4_NW8=0" "X=0"; do echo -n "300k $sw: "; env $sw timeout -s KILL 300 python tools/r05/proto_fit.py 150 300000 6 2>&1 | tail -2 | head -1 | cut -c1-230; done
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "protocol_shapes or exvivo" 2>&1 | tail -2
