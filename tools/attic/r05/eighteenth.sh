#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r
mkdir -p $O
timeout 600 python -m pytest tests/test_signal.py -m gpu -x -q 2>&1 | tail -2
PREP_ORDERS=F python bench.py --model prep --steps 10 --warmup 3 > $O/prep_small.json 2> /dev/null
PREP_ORDERS=F PREP_SHAPE=64,64,40 PREP_DIRAVG_SHAPE=160,160,100 python bench.py --model prep --steps 10 --warmup 3 > $O/prep_large.json 2> /dev/null
python - <<'PY'
import json
for tag in ('small', 'large'):
    d = json.load(open('gpurun_out/r05r/prep_%s.json' % tag))
    v = d['layouts']['diravg_F']
    print(tag, v['voxels'], 'voxels  %.3f ms  %.0f GB/s' % (v['kernel_ms'], v['achieved_GBs']), v.get('bit_exact_vs_numpy'))
PY
