#!/bin/bash
# kernel + memory-copy timeline of the host-buffer NODDI fit (no counters): where the time between 34 ms of kernels and
# the wall clock of the call goes
set -u
R=$PWD; O=$R/gpurun_out/host_tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python tools/host_fit_run.py 1000000 ${1:-f64} > $O/plain.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -- python tools/host_fit_run.py 1000000 ${1:-f64} > $O/prof.log 2>&1
find $O/tr -name '*.csv' | head
python tools/host_timeline.py $O/tr > $O/timeline.txt 2>&1
tail -80 $O/timeline.txt
cat $O/plain.log
