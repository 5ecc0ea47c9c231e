#!/bin/bash
# per-kernel durations of the NODDI headline (1 M voxels) for library builds: bash tools/r04/trace1m.sh default <variant> ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  unset AMICO_AMD_LIB; [ $v != default ] && export AMICO_AMD_LIB=$PWD/variants/$v/libamico_amd.so
  O=gpurun_out/r04n_$v
  mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --voxels 1000000 --no-cpu-baseline --no-other-configs > $O/trace.log 2>&1
  python tools/rocpd_summary.py $O/trace/t_results.db > $O/kernels.txt 2>&1
  rm -rf $O/trace
  echo "== $v"; cut -c1-80,88-135,175-200 $O/kernels.txt | grep -v "build_\|rocclr\|at::native" | head -26
done
