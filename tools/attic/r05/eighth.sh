#!/bin/bash
# round 5, eighth GPU call: k_prep_stream4 walking 1 / 2 / 4 adjacent tiles per wavefront (directional average, 306 volumes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h
mkdir -p $O
timeout 600 python -m pytest tests/test_signal.py -m gpu -x -q 2>&1 | tail -2
for vl in 1 2 4; do
  export AMX_PREP_VL=$vl
  PREP_ORDERS=F python bench.py --model prep --steps 10 --warmup 3 > $O/prep_small_$vl.json 2> /dev/null
  PREP_ORDERS=F PREP_SHAPE=64,64,40 PREP_DIRAVG_SHAPE=160,160,100 python bench.py --model prep --steps 10 --warmup 3 > $O/prep_large_$vl.json 2> /dev/null
  python - <<PY
import json
for tag in ('small', 'large'):
    d = json.load(open('gpurun_out/r05h/prep_%s_$vl.json' % tag))
    v = d['layouts']['diravg_F']
    print('VL=$vl', tag, v['voxels'], 'voxels  %.3f ms  %.0f GB/s' % (v['kernel_ms'], v['achieved_GBs']), v.get('bit_exact_vs_numpy'))
PY
done
