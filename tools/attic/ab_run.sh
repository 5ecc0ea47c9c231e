# A/B bench of library variants: bash tools/ab_run.sh <variant|default> ...   (AMX_NO_PAIR=1 in the env = the wavefront-per-voxel kernels)
for v in "$@"; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/ab_$v.log 2>&1
  echo "== $v"; grep -o '"value": [0-9.]*\|"stage_ms": \[[^]]*\]\|"max_abs_dmap": [0-9.e-]*\|"rerun_voxels": [0-9]*' gpurun_out/ab_$v.log | tr '\n' ' '; echo; tail -2 gpurun_out/ab_$v.log | grep -i "error\|Traceback" 
done
