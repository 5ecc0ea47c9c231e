#!/bin/bash
# (needs tools/r06/gcert_nw8.patch applied and `bash tools/build_variant.sh gc8 -DAMX_GCERT_NW=8`)
# NNLS certificates with 8 wavefronts per workgroup, one workgroup per CU (variants/gc8: -DAMX_GCERT_NW=8): half as many orientations'
# Gram matrices live per XCD -- do the G_PP gathers hit the L2 then?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a23; mkdir -p $O
V=${1:-gc8}
for lib in "" variants/$V/libamico_amd.so; do
  echo "== lib '$lib'" | tee -a $O/ab.txt
  AMICO_AMD_LIB=$lib AB_STEPS=12 timeout -s KILL 300 python tools/r06/fork_ab.py "100000 300000 1000000 4000000" "AMX_FORK=0" 2>&1 | grep voxels | tee -a $O/ab.txt
done
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
for lib in main $V; do
  L=""; [ $lib = $V ] && L=variants/$V/libamico_amd.so
  AMICO_AMD_LIB=$L timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_$lib -- $B > $O/pmc_$lib.log 2>&1
  python - $O/pmc_$lib $lib <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(list); c=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        c[r['Kernel_Name'][:60]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in d.items():
    if 'gcert' in k:
        m=len(v); cc=c[k]
        print('%-6s %-52s avg %.1f us  DRAM read %.0f MB  L2 hit %.2f'%(sys.argv[2],k[:52],sum(v)/m,cc['TCC_EA0_RDREQ_DRAM_32B_sum']*32/m/1e6,cc['TCC_HIT_sum']/max(1,cc['TCC_HIT_sum']+cc['TCC_MISS_sum'])))
PY
done 2>&1 | tee $O/kernels.txt
