#!/bin/bash
# the device pipeline with the tensor fit riding on the gather (one pass over the image) against gather + k_dti_dirs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "signal or pipeline or directions" > $O/gpu_tests_fused.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_fused.txt; grep -E "^FAILED|^E " $O/gpu_tests_fused.txt | head
for u in 1 0 1 0; do
  AMX_PIPELINE_UNFUSED=$u python bench.py --model pipeline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('unfused=$u: %.1f M voxels/s %.3f ms' % (d['value']/1e6, d['ms_per_step']), d['parity']['y_bit_exact'], d['parity']['frac_within_1e-6'], d['config'].get('gather_and_tensor_fit'))"
done 2>&1 | tee $O/pipeline_ab.txt
for u in 1 0; do
AMX_PIPELINE_UNFUSED=$u timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_pipe$u -o t -- python bench.py --model pipeline --steps 3 --warmup 1 > /dev/null 2>&1
python tools/rocpd_summary.py $O/kt_pipe$u/t_results.db 2>/dev/null | grep -E "k_prep_gather|k_dti|k_scatter" | cut -c1-130
rm -rf $O/kt_pipe$u
done 2>&1 | tee -a $O/pipeline_ab.txt
