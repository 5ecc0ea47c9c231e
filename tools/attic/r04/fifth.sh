#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "protocol or exvivo or golden" > $O/shape_tests.txt 2>&1; grep -n "passed\|failed\|Error" $O/shape_tests.txt | head
bash tools/r04/ab.sh "200000 1000000" default s1mp12 s1nw12 2>&1 | tee $O/ab_s1.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04e/bench_line.json'))
print('headline %.2f M voxels/s %.3f ms' % (d['value'] / 1e6, d['ms_per_step']), d['seed_chain'])
for k, v in d['other_configs'].items():
    print(k, '%.2f M voxels/s' % (v['value'] / 1e6), {a: v[a] for a in ('ms_per_step', 'ms_per_call', 'rate_per_byte_vs_headline', 'seed_chain', 'parity') if a in v})
PY
