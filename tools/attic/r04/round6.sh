#!/bin/bash
# exact-dual loop of the LASSO certificates: atoms per step (variants un1 = before, default 2 / 4 wide, w2 = 2 / 2, w2n3 = 3 / 2)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "50000 200000 1000000" un1 default w2 w2n3 un1 default 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "kkt" 2>&1 | grep "passed\|failed"
