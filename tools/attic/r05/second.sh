#!/bin/bash
# round 5, second GPU call: the shape-generality work (global-tile kernels, K-window table passes): new parity cases, suite, bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "other_protocol_shapes" > $O/shapes.txt 2>&1; tail -15 $O/shapes.txt
timeout 900 python -m pytest tests/test_gpu_solvers.py -m gpu -x -q > $O/solvers.txt 2>&1; tail -8 $O/solvers.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt
timeout 1200 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench_err.txt; tail -5 $O/bench_err.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05b/bench_line.json'))
print('headline %.1f M voxels/s' % (d['value'] / 1e6))
for k, v in d.get('other_configs', {}).items():
    print(k, '%.1f M voxels/s' % (v['value'] / 1e6), v.get('rate_per_byte_vs_headline'), v.get('seed_chain'), v.get('parity'))
PY
