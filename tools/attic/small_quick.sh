#!/bin/bash
# FreeWater 2M + SANDI 1M: kernel time, then all GPU tests of the small models
for m in freewater:2000000 sandi:1000000; do
python bench.py --model ${m%%:*} --voxels ${m##*:} --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit"\|"kernel_ms": [0-9.]*\|"max_abs_dmap": [0-9.e-]*' | head -3 | tr '\n' ' '; echo
done
python -m pytest tests/test_gpu_kkt.py tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_czb.py -m gpu -x -q 2>&1 | tail -3
