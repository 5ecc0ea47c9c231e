#!/bin/bash
# round 6, second GPU call: counter calibration (tools/r06/calib.sh) + kernel timelines of the forked fits (evidence for the negative result)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/r06/calib.sh > gpurun_out/r06_calib.log 2>&1; tail -120 gpurun_out/r06_calib.log
O=gpurun_out/r06_forktrace
mkdir -p $O
for f in 0 1 2; do
  AB_STEPS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/f$f -o f$f -- python tools/r06/fork_ab.py "100000 1000000" "AMX_FORK=$f" > $O/f$f.log 2>&1
  python tools/r06/fork_trace_summary.py $O/f$f | tee $O/timeline_fork$f.txt | tail -45
done
