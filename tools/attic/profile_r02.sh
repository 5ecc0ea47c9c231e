#!/bin/bash
# rocprofv3 evidence of round 2 (all single-GPU configs): bash tools/profile_r02.sh r02
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
bash tools/profile_round.sh $TAG > gpurun_out/profile_${TAG}_noddi.log 2>&1
bash tools/profile_small.sh $TAG > gpurun_out/profile_${TAG}_small.log 2>&1
# MFMA / VALU-mix counters of the NODDI stage kernels (separate passes)
O=gpurun_out/prof_$TAG
i=10
for set in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/pmc$i.log 2>&1
done
tail -2 gpurun_out/profile_${TAG}_noddi.log | cut -c1-300
