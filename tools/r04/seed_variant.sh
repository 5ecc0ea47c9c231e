#!/bin/bash
# variant of libamico_amd.so that differs in amx_seed.o only: tools/r04/seed_variant.sh <name> [-DFLAG ...]  (run `make` first)
set -e
name=$1; shift
root=$(cd $(dirname $0)/../.. && pwd)
out=$root/variants/$name
mkdir -p $out
cd $root/amico_amd/csrc
/opt/rocm/bin/hipcc -DAMX_S2_NW=16 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value "$@" -c -o $out/amx_seed.o amx_seed.hip 2>/dev/null
objs=""; for u in amx_api amx_noddi_s1 amx_noddi_s2 amx_noddi_s3 amx_fw amx_sandi amx_czb amx_small amx_signal amx_volume amx_batched amx_buildid; do objs="$objs $u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libamico_amd.so $out/amx_seed.o $objs
rm -f $out/amx_seed.o
ls -la $out/libamico_amd.so
