#!/bin/bash
# rescue pass of the NNLS certificates (corrected semi-normal equations for the supports refused for conditioning)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
timeout 1200 python -m pytest tests -m gpu -x -q -k "noddi or kkt or parity or fullsize" > gpurun_out/r04e/tests.txt 2>&1; tail -5 gpurun_out/r04e/tests.txt
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default 2>&1 | tee gpurun_out/r04e/ab.txt
AMX_NO_RESCUE=1 bash tools/r04/ab.sh "200000 1000000 4000000" default 2>&1 | sed 's/^default/norescue/' | tee -a gpurun_out/r04e/ab.txt
bash tools/r04/trace1m.sh default 2>&1 | head -22
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['solver_stats']); print(d.get('seed_chain')); print(d['parity'])"
bash tools/r04/stats1m.sh 2>&1 | cut -c1-400
