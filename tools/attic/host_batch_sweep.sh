#!/bin/bash
for t in f64 f32; do for n in 1000000 600000 2500000; do
  echo "== $t n $n"; python tools/host_fit_run.py $n $t 2>&1 | grep call | tail -3 | tr '\n' ' '; echo
done; done
python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
