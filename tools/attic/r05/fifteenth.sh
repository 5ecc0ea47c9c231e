#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05o
SECONDS=0; python bench.py > gpurun_out/r05o/bench_line.json 2> gpurun_out/r05o/bench_err.txt; echo "elapsed $SECONDS s"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05o/bench_line.json'))
r = d['roofline']
print('value %.1f M  ms %.3f  frac %.4f traffic %s whole %s  build %s' % (d['value']/1e6, d['ms_per_step'], r['frac'], r['traffic'], r['traffic_whole_fit'], d['build']))
print(d['compute_side'])
PY
