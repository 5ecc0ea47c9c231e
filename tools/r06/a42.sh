#!/bin/bash
# rocprofv3 kernel timeline of a 100 000-voxel fit after the stream's idle nodes went (compare gpurun_out/r06_forktrace/timeline_fork0.txt of the first session)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a42; mkdir -p $O
AB_STEPS=2 AB_PROFILING=0 timeout -s KILL 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python tools/r06/fork_ab.py "100000" "AMX_FORK=0" > $O/t.log 2>&1
python tools/r06/fork_trace_summary.py $O/t | tee $O/timeline.txt | tail -40
