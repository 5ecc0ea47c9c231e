#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/trace1m.sh default 2>&1 | head -16
bash tools/r04/ab.sh "200000 1000000 4000000" default 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "kkt or parity" 2>&1 | tail -3
