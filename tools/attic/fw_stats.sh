#!/bin/bash
# per-kernel times of the FreeWater (or sandi) bench: tools/fw_stats.sh [freewater|sandi] [voxels]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
M=${1:-freewater}; V=${2:-2000000}
rm -rf gpurun_out/fws; mkdir -p gpurun_out/fws
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fws -- python bench.py --model $M --voxels $V --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/fws.log 2>&1
f=$(find gpurun_out/fws -name '*kernel_stats.csv' | head -1)
python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-70s calls %4s avg %10.1f us  %5s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
P
