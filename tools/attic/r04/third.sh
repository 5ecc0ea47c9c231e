#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c
mkdir -p $O
bash tools/r04/ab.sh "50000 200000 1000000 4000000" default s1o2 s2o2 so2 2>&1 | tee $O/ab_occ.txt
AMICO_AMD_LIB=$PWD/variants/so2/libamico_amd.so timeout 600 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_multi.py -m gpu -x -q -k "noddi" > $O/so2_tests.txt 2>&1; tail -3 $O/so2_tests.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
