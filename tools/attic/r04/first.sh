#!/bin/bash
# round 4, first GPU call: the suite with the new tests, the default bench line, per-kernel traces over the call size (where the
# ~2.5 ms floor of the chain sits), WRITE_SIZE calibration against a known-size fill.   bash tools/r04/first.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
tail -5 $O/gpu_tests.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; tail -c 600 $O/bench_line.json
for v in 50000 200000 1000000; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$v -o t -- python bench.py --steps 5 --warmup 2 --voxels $v --no-cpu-baseline --no-other-configs > $O/trace_$v.log 2>&1
  python tools/rocpd_summary.py $O/trace_$v/t_results.db > $O/kernels_$v.txt 2>&1
done
# WRITE_SIZE / FETCH_SIZE against known sizes: a 1 GiB fill (writes only) and a 1 GiB copy (1 GiB read + 1 GiB written)
cat > /tmp/calib.py <<'PY'
import torch
a = torch.empty(1 << 28, dtype=torch.float32, device='cuda'); b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(3): a.fill_(1.5)
torch.cuda.synchronize()
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
PY
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/calib_$c -- python /tmp/calib.py > $O/calib_$c.log 2>&1
done
python - <<'PY' > gpurun_out/r04a/calibration.txt 2>&1
import csv, glob, collections
for c in ('WRITE_SIZE', 'FETCH_SIZE'):
    acc = collections.defaultdict(list)
    for fn in glob.glob('gpurun_out/r04a/calib_%s/*/*_counter_collection.csv' % c):
        for r in csv.DictReader(open(fn)):
            if r['Counter_Name'] == c: acc[r['Kernel_Name'][:90]].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(c, 'KiB mean %.6g n=%d  (1 GiB = 1048576 KiB)' % (sum(v) / len(v), len(v)), k)
PY
cat $O/calibration.txt
