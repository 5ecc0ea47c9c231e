"""Average the rocprofv3 --pmc counters (csv) of the kernels whose name contains a pattern.
usage: python tools/pmc_kernel.py <pattern> <counter_collection.csv> [...]"""
import collections
import csv
import sys

pat = sys.argv[1]
for f in sys.argv[2:]:
    acc = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = (r['Grid_Size'], r['Workgroup_Size'], r['LDS_Block_Size'], r['VGPR_Count'])
    print(f, 'grid/wg/lds/vgpr =', meta)
    for k, v in sorted(acc.items()):
        print('  %-24s n=%d mean=%.4g' % (k, len(v), sum(v) / len(v)))
