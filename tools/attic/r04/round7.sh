#!/bin/bash
# stage-3 seed solver: candidate byte lists through LDS (default) against the tree before (variants/prev)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "50000 200000 1000000 4000000" prev default prev default 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "kkt or parity or multi" 2>&1 | grep "passed\|failed"
