cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PREP_ORDERS=${1:-F}
mkdir -p gpurun_out/prep1 gpurun_out/prep2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prep1 -- python bench.py --model prep --steps 4 --warmup 1 > gpurun_out/prep1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prep2 -- python bench.py --model prep --steps 4 --warmup 1 > gpurun_out/prep2.log 2>&1
tail -2 gpurun_out/prep2.log | cut -c1-300
