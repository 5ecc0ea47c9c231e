#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_lut; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python bench.py --model lut --steps 3 --warmup 1 > $O/p$i.log 2>&1
done
python tools/pmc_table.py $O "k_lut"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/st -o st -- python bench.py --model lut --steps 5 --warmup 1 > $O/st.log 2>&1
python tools/rocpd_summary.py $O/st/st_results.db | cut -c1-170 | head -6
