cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mix
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d gpurun_out/mix -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/mix.log 2>&1
