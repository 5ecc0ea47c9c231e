#!/bin/bash
# round 5, fifth GPU call: the rescue pass of the NNLS certificates with its samples prefetched deeper -- does it pay at every size now?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e
mkdir -p $O
for rf in 2000000 0; do
  export AMX_RESCUE_FROM=$rf
  echo "== AMX_RESCUE_FROM=$rf"
  bash tools/r04/ab.sh "50000 200000 1000000 2000000" default ah2 ah1 2>&1 | tee -a $O/ab_rescue.txt
done
unset AMICO_AMD_LIB
export AMX_RESCUE_FROM=0
for v in default ah1; do
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/k_$v -o k -- python tools/r05/proto_fit.py bench 1000000 3 > $O/k_$v.log 2>&1
  python tools/rocpd_summary.py $O/k_$v/k_results.db | grep "gcert\|k_noddi<" | cut -c1-130
done
