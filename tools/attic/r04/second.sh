#!/bin/bash
# round 4, second GPU call: stage-2 products derived from the stage-1 table (one pass over y)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py -m gpu -x -q -s -k "noddi or protocol or chain or boundary or float32 or progress or lambda" > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
tail -4 $O/gpu_tests.txt
timeout 600 python tools/r04/s2_ab.py 300000 > $O/s2_ab.txt 2>&1; cat $O/s2_ab.txt
for v in 200000 1000000; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$v -o t -- python bench.py --steps 5 --warmup 2 --voxels $v --no-cpu-baseline --no-other-configs > $O/trace_$v.log 2>&1
  python tools/rocpd_summary.py $O/trace_$v/t_results.db > $O/kernels_$v.txt 2>&1
  grep '^{' $O/trace_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v voxels: %.2f M voxels/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']), d['seed_chain'])"
done
AMX_S2_EXACT=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exact pass for all voxels: %.2f M voxels/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']), d['seed_chain'])"
cut -c1-80,88-135 $O/kernels_1000000.txt | grep -v "build_\|rocclr\|at::native" | head -24
