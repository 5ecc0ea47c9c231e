# A/B of library variants on the NODDI headline (stage kernel times + seed kernels): bash tools/seed_ab.sh default nw12 ...
for v in "$@"; do
  unset AMICO_AMD_LIB
  [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  rm -rf gpurun_out/ab_$v; rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$v -o sp -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/ab_$v.log 2>&1
  echo "== $v"; python tools/rocpd_summary.py gpurun_out/ab_$v/sp_results.db | grep "k_noddi\|k_nnls\|k_lasso" | awk '{printf "%s %s %s %s | avg %.3f ms scratch %s\n", $2,$3,$4,$5, $(NF-10)/1e6, $(NF-2)}' | cut -c1-150
  grep -o '"value": [0-9.]*\|"max_abs_dmap": [0-9.e-]*' gpurun_out/ab_$v.log | tr '\n' ' '; echo
done
