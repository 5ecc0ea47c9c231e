cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
M=${1:-freewater}
rm -rf gpurun_out/fwp; mkdir -p gpurun_out/fwp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/fwp -- python bench.py --model $M --voxels 2000000 --steps 3 --warmup 1 > gpurun_out/fwp.log 2>&1
tail -1 gpurun_out/fwp.log | cut -c1-400
