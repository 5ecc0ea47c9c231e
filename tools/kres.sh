#!/bin/bash
# kernel resource usage of one csrc unit: tools/kres.sh amx_noddi_s3 [extra flags]
u=$1; shift
cd $(dirname $0)/../amico_amd/csrc
/opt/rocm/bin/hipcc -DAMX_S2_NW=16 -O3 -std=c++17 --offload-arch=gfx950 "$@" -c $u.hip -o /tmp/kres_$u.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
  awk '/Function Name/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)} / VGPRs:/{v=$4} /ScratchSize/{s=$5} /Occupancy/{o=$5} /VGPRs Spill/{print n, "vgpr="v, "scratch="s, "occ="o, "spill="$5}' | c++filt | cut -c1-150
