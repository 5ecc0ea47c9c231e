#!/bin/bash
# SeedFeed: a block's records touched when it is opened, next ticket requested ahead (variants/notouch: without)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default notouch default notouch 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "kkt or parity" 2>&1 | tail -3
bash tools/r04/trace1m.sh default 2>&1 | head -8
