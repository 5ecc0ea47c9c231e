#!/bin/bash
# round 5, sixth GPU call: the bandwidth rows at size: k_prep_gather / k_prep_stream4 on ~1.5 M masked voxels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f
mkdir -p $O
python bench.py --model prep --steps 10 --warmup 3 > $O/prep_small.json 2> $O/prep_small.err
PREP_SHAPE=192,192,120 PREP_DIRAVG_SHAPE=160,160,100 python bench.py --model prep --steps 10 --warmup 3 > $O/prep_large.json 2> $O/prep_large.err
python - <<'PY'
import json
for tag in ('small', 'large'):
    try:
        d = json.load(open('gpurun_out/r05f/prep_%s.json' % tag))
    except Exception as e:
        print(tag, 'failed', e); continue
    for k, v in d['layouts'].items():
        print(tag, k, v['voxels'], 'voxels  %.3f ms  %.0f GB/s' % (v['kernel_ms'], v['achieved_GBs']), 'f32 rows: %.3f ms %.0f GB/s' % (v['float32_rows']['kernel_ms'], v['float32_rows']['achieved_GBs']) if 'float32_rows' in v else '', v.get('bit_exact_vs_numpy'))
PY
tail -3 $O/prep_large.err
python bench.py --model dti --steps 10 --warmup 3 2>/dev/null | cut -c1-600
