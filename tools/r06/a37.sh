#!/bin/bash
# the mask gather's tile in parts (twice / three times the wavefronts per CU): bit-exactness tests, then the bandwidth rows
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a37; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_signal.py tests/test_gpu_boundary.py -x -q -m gpu > $O/tests.txt 2>&1; grep -n "passed\|failed" $O/tests.txt | tail -2
for rep in 1 2; do
for parts in 1 2 3 4; do
  echo -n "rep $rep AMX_PREP_PARTS=$parts: " | tee -a $O/ab.txt
  AMX_PREP_PARTS=$parts timeout -s KILL 300 python bench.py --model prep --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
def walk(o, pre=''):
    out = []
    if isinstance(o, dict):
        for k, v in o.items():
            if isinstance(v, (dict, list)): out += walk(v, pre + k + '.')
            elif isinstance(v, (int, float)) and ('ms' in k or 'GBs' in k or 'gbs' in k.lower() or k == 'value'): out.append('%s%s=%.4g' % (pre, k, v))
    return out
print(' '.join(walk(d))[:900])" | tee -a $O/ab.txt
done
done
