#!/bin/bash
# round 5, seventeenth GPU call: what do the bandwidth kernels really move?  FETCH_SIZE / WRITE_SIZE / L2 hit counters of k_prep_stream4 and k_prep_gather at size
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q
mkdir -p $O
export PREP_ORDERS=F PREP_SHAPE=192,192,120 PREP_DIRAVG_SHAPE=160,160,100
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python bench.py --model prep --steps 3 --warmup 1 > $O/pmc$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('gpurun_out/r05q/pmc*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if 'k_prep' in k:
            acc[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print('   %-32s n=%d mean=%.6g' % (n, len(v), sum(v) / len(v)))
PY
