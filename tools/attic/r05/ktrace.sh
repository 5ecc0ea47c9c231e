#!/bin/bash
# kernel stats of one bench leg: bash tools/r05/ktrace.sh <tag> [bench args]   -> gpurun_out/<tag>_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/kt_$TAG
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $O/bench.log 2>&1
python tools/rocpd_summary.py $O/t/t_results.db > gpurun_out/${TAG}_kernel_stats.txt 2>&1
grep '^{' $O/bench.log | tail -1 | cut -c1-200 >> gpurun_out/${TAG}_kernel_stats.txt
rm -rf $O/t
head -24 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-150
