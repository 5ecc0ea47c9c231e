#!/bin/bash
# the small models after the serial-load fixes (SANDI maps section, CylinderZeppelinBall pivoting loop) against variants/base
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
for v in base default; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  for m in sandi czb freewater; do
    python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v $m: %.1f M voxels/s  %.3f ms  kernel %s' % (d['value']/1e6, d['ms_per_step'], d.get('roofline',{}).get('kernel_ms')), d.get('parity'))"
  done
done 2>&1 | tee $O/small_ab.txt
unset AMICO_AMD_LIB
timeout 900 python -m pytest tests -m gpu -x -q -k "czb or sandi or small or fullsize" > $O/gpu_tests_small.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_small.txt
