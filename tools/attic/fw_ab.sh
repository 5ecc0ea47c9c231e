#!/bin/bash
# A/B of FreeWater variants: tools/fw_ab.sh <variant>...
for v in "" "$@"; do
  echo "== ${v:-default}"
  if [ -n "$v" ]; then export AMICO_AMD_LIB=variants/$v/libamico_amd.so; else unset AMICO_AMD_LIB; fi
  bash tools/fw_stats.sh freewater | head -2
done
