#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_czb.py tests/test_gpu_solvers.py -m gpu -x -q > $O/new_tests.txt 2>&1; grep -n "passed\|failed\|Error\|assert" $O/new_tests.txt | head -20
timeout 600 python bench.py --model czb --voxels 500000 --steps 5 --warmup 2 --no-cpu-baseline > $O/czb.json 2> $O/czb.err; python -c "
import json; d=json.load(open('$O/czb.json')); print('czb %.1f M voxels/s %.3f ms kernel %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']), d['parity'], d['solver_stats'])"; tail -3 $O/czb.err
bash tools/r04/ab.sh "100000 200000 300000 1000000" default 2>&1 | tee $O/ab.txt
