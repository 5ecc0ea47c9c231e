#!/bin/bash
# variant of libamico_amd.so that differs in ONE unit: tools/r04/unit_variant.sh <name> <unit> [-DFLAG ...]  (run `make` first)
set -e
name=$1; unit=$2; shift; shift
root=$(cd $(dirname $0)/../.. && pwd)
out=$root/variants/$name
mkdir -p $out
cd $root/amico_amd/csrc
/opt/rocm/bin/hipcc -DAMX_S2_NW=16 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value "$@" -c -o $out/$unit.o $unit.hip 2>/dev/null
objs=""; for u in amx_api amx_seed amx_noddi_s1 amx_noddi_s2 amx_noddi_s3 amx_fw amx_sandi amx_czb amx_small amx_signal amx_volume amx_batched amx_buildid; do [ $u != $unit ] && objs="$objs $u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libamico_amd.so $out/$unit.o $objs
rm -f $out/$unit.o
ls -la $out/libamico_amd.so
