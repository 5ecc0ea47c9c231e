#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fwg; mkdir -p gpurun_out/fwg
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fwg -- python tools/fw_gather_probe.py 2>&1 | grep "per fit"
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/fwg/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_fw_project' in r['Kernel_Name'] or 'k_freewater_refill' in r['Kernel_Name']]
for name in ('k_fw_project', 'k_freewater_refill'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if name in r['Kernel_Name']]
    print(name, 'first config: %.1f us   second config: %.1f us' % (sum(d[3:13]) / 10, sum(d[16:26]) / 10))
P
