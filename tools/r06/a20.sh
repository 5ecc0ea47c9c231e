#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a20; mkdir -p $O
timeout -s KILL 1500 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; tail -c 1500 $O/bench_line.json; echo
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -o /tmp/counter_probe tools/probes/counter_probe.hip && /tmp/counter_probe 2 3 | tee $O/probe_planes.txt
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/planes -- /tmp/counter_probe 2 1 > $O/planes.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r06_a20/planes/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_read_planes' in r['Kernel_Name']:
            print(r['Counter_Name'], r['Counter_Value'])
PY
