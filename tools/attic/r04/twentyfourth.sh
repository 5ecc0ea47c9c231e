#!/bin/bash
# nine scan tiles + the lone iso atom by hand in k_nnls_seed<1> (variants/prev: ten tiles)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default prev default prev 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "kkt or parity" 2>&1 | tail -3
