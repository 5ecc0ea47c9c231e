#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
AMX_DEBUG=1 timeout -s KILL 600 python tools/r06/big_time.py 50000 2>&1 | grep "^lambda\|lasso_big:" | head -20
timeout -s KILL 600 python -m pytest tests/test_gpu_solvers.py -m gpu -q -k "dense" 2>&1 | tail -2
