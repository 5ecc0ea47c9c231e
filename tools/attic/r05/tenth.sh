#!/bin/bash
# round 5, tenth GPU call: FreeWater in one kernel (producer / consumer wavefronts) -- parity first (short time-outs: a hang must not cost the box), then A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j
mkdir -p $O
timeout 180 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "freewater or small_models or golden" > $O/fw_tests1.txt 2>&1; echo "rc=$?"; tail -5 $O/fw_tests1.txt
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kkt.py tests/test_gpu_boundary.py tests/test_gpu_multi.py -m gpu -x -q -k "freewater or FreeWater or fw or other_protocol or small" > $O/fw_tests2.txt 2>&1; echo "rc=$?"; tail -5 $O/fw_tests2.txt
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "freewater" > $O/fw_tests3.txt 2>&1; echo "rc=$?"; tail -3 $O/fw_tests3.txt
for nf in 0 1; do
  export AMX_FW_NO_FUSE=$nf
  timeout 300 python bench.py --model freewater --voxels 2000000 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('NO_FUSE=$nf: %.1f M voxels/s  %.3f ms/step  kernels %.3f ms  frac %.3f  dmap %.1e' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms'], r['frac'], d['parity']['max_abs_dmap']), d.get('float32_signals_in_hbm', {}).get('kernel_ms'))"
done
