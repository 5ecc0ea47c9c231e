#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_multi.py tests/test_gpu_parity.py -m gpu -x -q -k "noddi or protocol" > $O/tests.txt 2>&1; grep -n "passed\|failed\|Error\|assert" $O/tests.txt | head
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default 2>&1 | tee $O/ab.txt
AMX_GCERT_GLOBAL=1 bash tools/r04/ab.sh "200000 1000000" default 2>&1 | tee -a $O/ab.txt
for v in 1000000; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$v -o t -- python bench.py --steps 5 --warmup 2 --voxels $v --no-cpu-baseline --no-other-configs > $O/trace_$v.log 2>&1
  python tools/rocpd_summary.py $O/trace_$v/t_results.db > $O/kernels_$v.txt 2>&1
  cut -c1-80,88-135 $O/kernels_$v.txt | grep "gcert"
done
