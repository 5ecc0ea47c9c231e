#!/bin/bash
# A/B builds of libamico_amd.so with extra compile flags: tools/build_variant.sh <name> [-DFLAG ...]
# -> variants/<name>/libamico_amd.so (git-ignored, travels to the GPU box); select it with AMICO_AMD_LIB=variants/<name>/libamico_amd.so
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
out=$root/variants/$name
mkdir -p $out
cd $root/amico_amd/csrc
FLAGS="-DAMX_S2_NW=16 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value"
pids=()
for u in amx_api amx_big amx_seed amx_noddi_s1 amx_noddi_s2 amx_noddi_s3 amx_fw amx_sandi amx_czb amx_small amx_signal amx_volume amx_batched amx_buildid; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $out/$u.o $u.hip 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libamico_amd.so $out/*.o
rm -f $out/*.o
ls -la $out/libamico_amd.so
