#!/bin/bash
# round 5, eleventh GPU call: fused FreeWater with 7 / 5 / 3 tiles in flight, chunk sizes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m
mkdir -p $O
timeout 180 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "freewater or small_models or golden" > $O/fw_tests1.txt 2>&1; echo "rc=$?"; tail -3 $O/fw_tests1.txt
run() {
  timeout 300 python bench.py --model freewater --voxels 2000000 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1: %.1f M voxels/s  %.3f ms/step  kernels %.3f ms  frac %.3f  dmap %.1e  f32 kernels' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms'], r['frac'], d['parity']['max_abs_dmap']), d.get('float32_signals_in_hbm', {}).get('kernel_ms'))"
}
AMX_FW_NO_FUSE=1 run "pair"
for v in default pw0 pw1 pw3; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  run "fused $v chunk default"
  AMX_REFILL_CHUNK=2048 run "fused $v chunk 2048"
done
