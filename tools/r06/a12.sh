#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for n in 300000 1000000; do
  for sw in "AMX_GCERT2_THIRD=0" "AMX_GCERT2_THIRD=1" "X=0" "AMX_GCERT2_THIRD_MIN=16"; do
    echo -n "rep $rep $sw: "
    env $sw timeout -s KILL 200 python tools/r05/proto_fit.py 105 $n 8 2>&1 | tail -2 | head -1 | cut -c1-200
  done
done
done
