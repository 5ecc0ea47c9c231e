#!/bin/bash
# (AMX_DBG_GRAM0 is part of tools/r06/table_vm.patch, not of the tree)
# experiment: how much of the NNLS certificates' time and DRAM traffic is the per-orientation Gram matrix missing the L2?  (AMX_DBG_GRAM0=1: all chunks read matrix 0)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a22; mkdir -p $O
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
for v in 0 1; do
  AMX_DBG_GRAM0=$v timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_$v -- $B > $O/pmc_$v.log 2>&1
  python - $O/pmc_$v $v <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(list); c=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
disp={}
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        c[r['Kernel_Name'][:60]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in d.items():
    if 'gcert' in k:
        m=len(v); cc=c[k]
        print('GRAM0=%s %-52s avg %.1f us  DRAM read %.0f MB  L2 hit %.2f'%(sys.argv[2],k[:52],sum(v)/m,cc['TCC_EA0_RDREQ_DRAM_32B_sum']*32/m/1e6,cc['TCC_HIT_sum']/max(1,cc['TCC_HIT_sum']+cc['TCC_MISS_sum'])))
PY
done 2>&1 | tee $O/result.txt
