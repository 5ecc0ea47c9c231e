#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l
mkdir -p $O
timeout 600 python -m pytest tests/test_signal.py -m gpu -x -q > $O/tests.txt 2>&1; grep -n "passed\|failed\|Error\|assert" $O/tests.txt | head
for e in 0 1; do
  AMX_PREP_SCALAR=$e python bench.py --model prep --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); L=d['layouts']
print('AMX_PREP_SCALAR=$e', {k:(round(v['kernel_ms'],4), round(v['achieved_GBs'])) for k,v in L.items()}, 'diravg exact', L['diravg_F']['bit_exact_vs_numpy'], 'f32 rows', {k:(round(v['float32_rows']['kernel_ms'],4), round(v['float32_rows']['achieved_GBs'])) for k,v in L.items() if 'float32_rows' in v})"
done
