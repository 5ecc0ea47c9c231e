#!/bin/bash
# kernel timelines (rocprofv3 --kernel-trace) of the tree's library against the round-5 build at 100 000 and 1 M voxels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a06
mkdir -p $O
for v in cur r05; do
  unset AMICO_AMD_LIB
  [ $v = r05 ] && export AMICO_AMD_LIB=$PWD/variants/r05/libamico_amd.so
  AB_STEPS=4 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/$v -o $v -- python tools/r06/fork_ab.py "100000 1000000" "AMX_FORK=0" > $O/$v.log 2>&1
  grep "^AMX" $O/$v.log
  python tools/r06/fork_trace_summary.py $O/$v > $O/timeline_$v.txt
done
unset AMICO_AMD_LIB
paste <(grep "k_" $O/timeline_cur.txt | awk '{print $1, $NF}' ) <(grep "k_" $O/timeline_r05.txt | awk '{print $NF}') | column -t | head -60
python tools/r06/fork_ab.py "50000 100000 200000 300000 1000000" "AMX_FORK=0" | grep "^AMX"
