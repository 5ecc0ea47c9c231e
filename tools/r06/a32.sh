#!/bin/bash
# after the r06b profiles are summarised (pmc_traffic.json of this build): the bench line with its counter figures, stress parity
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a32; mkdir -p $O
timeout -s KILL 1500 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt; tail -c 600 $O/bench_line.json; echo
bash tools/r06/stress.sh
timeout -s KILL 1500 python tools/r06/stress_protocols.py 1000000 5 11 2>&1 | grep -v amdgpu.ids | tee $O/stress_protocols.txt | cut -c1-160
