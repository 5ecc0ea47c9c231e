#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for f in 0 1 2; do
AMX_FLOW=$f timeout 600 bash tools/r04/ab.sh "50000 200000 1000000" default 2>&1 | sed "s/^default/flow=$f /"
done
AMX_FLOW=1 timeout 900 python -m pytest tests -m gpu -x -q -k "hard_mix" 2>&1 | tail -30
