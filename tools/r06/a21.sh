#!/bin/bash
# (needs tools/r06/table_vm.patch applied: the tree keeps the row-major table; variants/tabrm = the same sources with -DAMX_TAB_VM=0)
# the A'y table with voxel-major atom tiles against the row-major one (variants/tabrm): parity, fit times, per-kernel times, DRAM counters
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a21; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kkt.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
for lib in "" variants/tabrm/libamico_amd.so; do
  echo "== lib '$lib'" | tee -a $O/ab.txt
  AMICO_AMD_LIB=$lib AB_STEPS=12 timeout -s KILL 300 python tools/r06/fork_ab.py "100000 300000 1000000 4000000" "AMX_FORK=0" 2>&1 | grep voxels | tee -a $O/ab.txt
done
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
for lib in new tabrm; do
  L=""; [ $lib = tabrm ] && L=variants/tabrm/libamico_amd.so
  AMICO_AMD_LIB=$L timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/ks_$lib -o ks -- $B > $O/ks_$lib.log 2>&1
  AMICO_AMD_LIB=$L timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum --output-format csv -d $O/pmc_$lib -- $B > $O/pmc_$lib.log 2>&1
  python - $O $lib <<'PY'
import csv, glob, sys, collections
O, lib = sys.argv[1], sys.argv[2]
for f in glob.glob(O + '/ks_' + lib + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:16]:
        print('%-64s calls %4s avg %9.1f us' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3))
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob(O + '/pmc_' + lib + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:64]
        a = acc[k]
        if r['Counter_Name'].startswith('TCC_EA0_RDREQ'): a[1] += float(r['Counter_Value']) * 32; a[0] += 1
        else: a[2] += float(r['Counter_Value']) * 32
for k, a in sorted(acc.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:12]:
    n = max(a[0], 1)
    print('DRAM %-64s read %8.1f MB written %8.1f MB per launch' % (k, a[1] / n / 1e6, a[2] / n / 1e6))
PY
done 2>&1 | tee $O/kernels.txt
