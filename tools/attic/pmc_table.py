"""Per-kernel table of rocprofv3 --pmc counters (mean per launch): python tools/pmc_table.py <dir> <kernel-name pattern>"""
import collections, csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    if max(len(v) for v in acc[k].values()) == 0:
        continue
    print(k[:110])
    for c, v in sorted(acc[k].items()):
        print('    %-28s n=%d mean=%.5g' % (c, len(v), sum(v) / len(v)))
