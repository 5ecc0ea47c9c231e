#!/bin/bash
# host-buffer NODDI fit: two compute streams vs one (AMX_HOST_ONE_STREAM=1), f64 and f32 signals
for t in f64 f32; do
  echo "== $t two streams"; python tools/host_fit_run.py 1000000 $t 2>&1 | grep call
  echo "== $t one stream";  AMX_HOST_ONE_STREAM=1 python tools/host_fit_run.py 1000000 $t 2>&1 | grep call
done
python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
