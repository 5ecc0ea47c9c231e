#!/bin/bash
# (AMX_RESCUE_TILE_GLOBAL was a three-line knob for this run only -- `g.tile_in_lds = fits && !knob` in amx_launch_noddi_gcert --, not kept: profiles/r06_protocols_ab.txt)
# the rescue pass without the tile staged in LDS (atoms of a support from the L2-resident tile): does it pay below 2 M voxels then?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a30; mkdir -p $O
for snr in 30 50 10; do
  echo "== SNR $snr" | tee -a $O/ab.txt
  AB_SNR=$snr AB_PROFILING=0 AB_STEPS=12 timeout -s KILL 600 python tools/r06/fork_ab.py "200000 300000 1000000" "AMX_FORK=0" "AMX_RESCUE_FROM=0" "AMX_RESCUE_FROM=0 AMX_RESCUE_TILE_GLOBAL=1" 2>&1 | grep voxels | tee -a $O/ab.txt
done
