#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
for m in 20 24 26 30; do
  echo "== AMX_SEED2_MAXATOMS=$m"
  AMX_SEED2_MAXATOMS=$m python tools/r05/proto_fit.py hcp 1000000 4 2>/dev/null | cut -c1-330
done 2>&1 | tee $O/seed2_maxatoms.txt
echo "== auto"; python tools/r05/proto_fit.py hcp 1000000 4 2>/dev/null | head -1 | tee -a $O/seed2_maxatoms.txt
python tools/r05/proto_fit.py bench 1000000 4 2>/dev/null | head -1
timeout 900 python -m pytest tests -m gpu -x -q -k "protocol_shapes" 2>&1 | tail -2
