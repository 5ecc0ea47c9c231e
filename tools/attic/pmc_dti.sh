#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_dti; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python bench.py --model dti --steps 2 --warmup 1 > $O/p$i.log 2>&1
done
python - <<'P'
import csv, glob, collections
acc = collections.defaultdict(list); dur = []
for fn in glob.glob('gpurun_out/pmc_dti/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        if 'k_dti_dirs' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(acc.items()):
    print('%-28s n=%d mean=%.6g' % (c, len(v), sum(v) / len(v)))
P
grep -h '^{' gpurun_out/pmc_dti/p1.log | cut -c1-500
