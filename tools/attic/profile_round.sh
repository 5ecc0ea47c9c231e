#!/bin/bash
# rocprofv3 evidence for one round: kernel stats of bench.py (NODDI, dti, prep) + PMC passes of the NODDI bench.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01d
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-rXX}
O=gpurun_out/prof_$TAG
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/noddi -o noddi -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $O/noddi_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/dti -o dti -- python bench.py --model dti --steps 5 --warmup 1 > $O/dti_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prep -o prep -- python bench.py --model prep --steps 5 --warmup 1 > $O/prep_bench.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $O/pmc$i.log 2>&1
done
for m in dti prep; do
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python bench.py --model $m --steps 3 --warmup 1 > $O/pmc$i.log 2>&1
  done
done
tail -1 $O/noddi_bench.log | cut -c1-300
find $O -name "*.db" -o -name "*stats*.csv" | head
