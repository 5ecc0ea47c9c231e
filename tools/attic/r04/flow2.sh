#!/bin/bash
# the seeded NODDI chain on two streams with chunk-level hand-over (AMX_FLOW=1) against one stream
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default 2>&1
AMX_FLOW=1 timeout 600 bash tools/r04/ab.sh "50000 200000 1000000 4000000" default 2>&1 | sed 's/^default/flow   /'
AMX_FLOW=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['solver_stats']); print(d.get('seed_chain')); print(d['parity'])"
AMX_FLOW=1 timeout 900 python -m pytest tests -m gpu -x -q -k "kkt or parity or skewed or repeatable" 2>&1 | tail -3
