cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 200 python bench.py --model dti --steps 10 --warmup 2 2>&1 | tail -1 | cut -c300-700
mkdir -p gpurun_out/dti1 gpurun_out/dti2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/dti1 -- python bench.py --model dti --steps 4 --warmup 1 > gpurun_out/dti1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/dti2 -- python bench.py --model dti --steps 4 --warmup 1 > gpurun_out/dti2.log 2>&1
ls gpurun_out/dti1/*/ | head
