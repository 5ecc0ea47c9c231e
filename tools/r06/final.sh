#!/bin/bash
# suite + default bench line + profiles of the tree as it is: bash tools/r06/final.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r06a}
O=gpurun_out/final_$TAG
mkdir -p $O
timeout -s KILL 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; tail -c 300 $O/bench_line.json
bash tools/r06/profile.sh $TAG
bash tools/profile_small.sh $TAG
for v in 50000 200000 500000 1000000 2000000 4000000 8000000; do
  python bench.py --steps 5 --warmup 2 --voxels $v --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v voxels: %.2f M voxels/s  %.2f ms   max |dmap| %.1e' % (d['value']/1e6, d['ms_per_step'], d['parity']['max_abs_dmap']))"
done | tee $O/size_scan.txt
