#!/usr/bin/env python3
"""counters of tools/probes/counter_probe.hip beside its known byte counts: python tools/r06/calib_summary.py gpurun_out/r06_calib"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
known = {}
for line in open(d + '/probe_plain.txt'):
    m = re.match(r'PROBE (\S+)\s+useful_bytes (\d+)\s+sector_bytes_64B (\d+)\s+launches (\d+)\s+best_ms ([\d.]+)', line)
    if m:
        known[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(5)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'^void ', '', r['Kernel_Name']).split('(')[0]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
print('kernel               useful MB  sectors(64B) MB   best ms | counter: mean per launch (and, for byte-like counters, the ratio to useful / sector bytes)')
for k, (useful, sect, ms) in known.items():
    print('%-20s %9.1f %12.1f %9.3f' % (k, useful / 1e6, sect / 1e6, ms))
    kk = [x for x in acc if x == k or x.startswith(k)]
    for name in kk:
        for c, v in sorted(acc[name].items()):
            m = sum(v) / len(v)
            extra = ''
            if c in ('FETCH_SIZE', 'WRITE_SIZE'):
                b = m * 1024
                extra = '  = %.1f MB: %.3f x useful, %.3f x sectors' % (b / 1e6, b / useful, b / sect)
            elif c.endswith('_32B_sum') and 'DRAM' in c:
                b = m * 32
                extra = '  x 32 B = %.1f MB: %.3f x useful, %.3f x sectors' % (b / 1e6, b / useful, b / sect)
            elif c in ('TCC_EA0_RDREQ_sum', 'TCC_EA0_WRREQ_sum', 'TCC_EA0_RDREQ_DRAM_sum', 'TCC_EA0_WRREQ_DRAM_sum'):
                extra = '  requests; bytes per request if all sectors moved once: %.1f' % (sect / m if m else 0)
            print('      %-34s n=%d %.6g%s' % (c, len(v), m, extra))
    a = acc.get(kk[0], {}) if kk else {}
    if 'TCC_EA0_RDREQ_128B_sum' in a:
        g = lambda c: sum(a.get(c, [0])) / max(1, len(a.get(c, [0])))
        r32, r64, r128, rall = g('TCC_EA0_RDREQ_32B_sum'), g('TCC_EA0_RDREQ_64B_sum'), g('TCC_EA0_RDREQ_128B_sum'), g('TCC_EA0_RDREQ_sum')
        b = 32 * r32 + 64 * r64 + 128 * r128
        print('      => 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B = %.1f MB: %.3f x useful, %.3f x sectors   (RDREQ_sum %.6g; 32B + 64B + 128B = %.6g)' % (
            b / 1e6, b / useful, b / sect, rall, r32 + r64 + r128))
