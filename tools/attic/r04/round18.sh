#!/bin/bash
# k_dti_dirs with the table-driven logarithm (default) against the tree before (variants/prev)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in prev default prev default; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  timeout 300 python bench.py --model dti --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', '%.1f M voxels/s' % (d['value']/1e6), 'kernel %.4f ms %.0f GB/s' % (r.get('kernel_ms', 0), r['achieved']), {k: d[k] for k in d if 'parity' in k or 'err' in k})"
done
unset AMICO_AMD_LIB
timeout 600 python -m pytest tests/test_signal.py tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | grep "passed\|failed"
