cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
AMICO_AMD_LIB=$PWD/variants/phases/libamico_amd.so AMX_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 0 --voxels 1000000 --no-cpu-baseline --no-other-configs > gpurun_out/phases.log 2>&1
grep -c "" gpurun_out/phases.log; grep "phases\|seeds:" gpurun_out/phases.log | tail -5; tail -3 gpurun_out/phases.log | cut -c1-300
