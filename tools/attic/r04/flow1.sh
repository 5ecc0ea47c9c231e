#!/bin/bash
# prototype: k_nnls_gcert<1> beside k_nnls_seed<1> on a second stream, chunk-level hand-over (AMX_FLOW=1)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r04/ab.sh "200000 1000000 4000000" default 2>&1
AMX_FLOW=1 timeout 300 bash tools/r04/ab.sh "200000 1000000 4000000" default 2>&1 | sed 's/^default/flow   /'
AMX_FLOW=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['solver_stats']); print(d.get('seed_chain')); print(d['parity'])"
