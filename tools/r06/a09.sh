#!/bin/bash
# other protocols / noise levels: third LASSO certificate pass, certificates' second look, rescue pass -- on or off
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a09
mkdir -p $O
for p in 105 150; do
  for sw in "X=0" "AMX_GCERT2_THIRD=1" "AMX_GCERT_REPAIR=1" "AMX_GCERT2_THIRD=1 AMX_GCERT_REPAIR=1" "AMX_GCERT2_THIRD=1 AMX_SEED2_MAXATOMS=26" "AMX_RESCUE_FROM=0"; do
    echo "== $p volumes, $sw"
    env $sw timeout -s KILL 200 python tools/r05/proto_fit.py $p 1000000 5 2>&1 | grep -v "^$" | tail -2 | cut -c1-330
  done
done | tee $O/protocols_ab.txt
for snr in 50 10; do
  for sw in "X=0" "AMX_RESCUE_FROM=0" "AMX_GCERT_REPAIR=1"; do
    echo "== SNR $snr, $sw"
    env $sw AB_SNR=$snr AB_STEPS=6 timeout -s KILL 200 python tools/r06/fork_ab.py "300000 1000000" "AMX_FORK=0" 2>&1 | grep "^AMX"
  done
done | tee $O/snr_ab.txt
