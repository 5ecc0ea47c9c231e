#!/bin/bash
# round 5, seventh GPU call: one memset for every list count of the chain; k_prep_stream4 without single-load loops; whole suite
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt
bash tools/r04/ab.sh "50000 200000 1000000" default 2>&1 | tee $O/ab.txt
PREP_ORDERS=F python bench.py --model prep --steps 10 --warmup 3 > $O/prep_small.json 2> $O/prep_small.err
PREP_ORDERS=F PREP_SHAPE=192,192,120 PREP_DIRAVG_SHAPE=160,160,100 python bench.py --model prep --steps 10 --warmup 3 > $O/prep_large.json 2> $O/prep_large.err
python - <<'PY'
import json
for tag in ('small', 'large'):
    try:
        d = json.load(open('gpurun_out/r05g/prep_%s.json' % tag))
    except Exception as e:
        print(tag, 'failed', e); continue
    for k, v in d['layouts'].items():
        print(tag, k, v['voxels'], 'voxels  %.3f ms  %.0f GB/s' % (v['kernel_ms'], v['achieved_GBs']), v.get('bit_exact_vs_numpy'))
PY
