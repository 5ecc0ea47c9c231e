#!/bin/bash
# rocprofv3 evidence for the configs besides the NODDI headline: FreeWater 2 M, SANDI 1 M, LUT resampling, signal preparation.
# usage (on the GPU box, from the repo root): bash tools/profile_small.sh r02a
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-rXX}
O=gpurun_out/small_$TAG
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/fw -o fw -- python bench.py --model freewater --voxels 2000000 --steps 5 --warmup 1 --no-cpu-baseline > $O/fw_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/sandi -o sandi -- python bench.py --model sandi --voxels 1000000 --steps 5 --warmup 1 --no-cpu-baseline > $O/sandi_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/czb -o czb -- python bench.py --model czb --voxels 500000 --steps 5 --warmup 1 --no-cpu-baseline > $O/czb_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/lut -o lut -- python bench.py --model lut --steps 5 --warmup 1 --no-cpu-baseline > $O/lut_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prep -o prep -- python bench.py --model prep --steps 5 --warmup 1 --no-cpu-baseline > $O/prep_bench.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  for m in freewater sandi czb; do
    v=1000000; [ $m = freewater ] && v=2000000; [ $m = czb ] && v=500000
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${m}_$i -- python bench.py --model $m --voxels $v --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_${m}_$i.log 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_lut_$i -- python bench.py --model lut --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_lut_$i.log 2>&1
  # the bandwidth rows (f2, f3): mask gather / directional average / scatter (round 6: their counters at size)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_prep_$i -- python bench.py --model prep --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_prep_$i.log 2>&1
done
# ... and two more sets for those rows only: where their wave cycles go (LDS issue stalls, the texture-address / data units)
i=20
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_prep_$i -- python bench.py --model prep --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_prep_$i.log 2>&1 || echo "prep pass $i failed"
done
for f in fw sandi czb lut prep; do tail -1 $O/${f}_bench.log | cut -c1-600; done
