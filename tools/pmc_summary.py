#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel (csv output).  usage: pmc_summary.py dir [dir ...]"""
import csv
import collections
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    if not k.startswith('void amx::k_noddi') and not k.startswith('void amx::k_'):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print('    %-28s n=%d mean=%.6g' % (c, len(v), sum(v) / len(v)))
