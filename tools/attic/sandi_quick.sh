#!/bin/bash
python bench.py --model sandi --voxels 1000000 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit"\|"kernel_ms": [0-9.]*\|"max_abs_dmap": [0-9.e-]*' | head -3 | tr '\n' ' '; echo
python -m pytest tests/test_gpu_kkt.py tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -x -q -k "sandi or SANDI or float32" 2>&1 | tail -3
