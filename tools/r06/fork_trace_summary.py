#!/usr/bin/env python3
"""timeline of the LAST fit of each size in a rocprofv3 --kernel-trace csv: start / end of every kernel relative to the fit's first kernel,
and which stream (queue) it ran on -- shows what a forked fit's side stream overlaps with.  usage: fork_trace_summary.py <dir>"""
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?') + '/' + r.get('Stream_Id', '?'), re.sub(r'^(void )?(amx::)?', '', r['Kernel_Name']).split('(')[0][:44]))
rows.sort()
# a fit starts with k_dir_to_lut
starts = [i for i, r in enumerate(rows) if r[3].startswith('k_dir_to_lut')]
fits = [(a, b) for a, b in zip(starts, starts[1:] + [len(rows)])]
# group fits by their number of gemm-kernel nanoseconds (a proxy of the size): print the last fit of each distinct launch sequence length
seen = {}
for a, b in fits:
    ks = rows[a:b]
    dur = ks[-1][1] - ks[0][0]
    gemm = sum(e - s for s, e, q, k in ks if k.startswith('k_noddi_gemm<false'))
    seen[round(gemm / 2e4)] = (a, b)      # (buckets of 20 us of GEMM time: one per call size)
for key in sorted(seen):
    a, b = seen[key]
    ks = [k for k in rows[a:b] if not k[3].startswith('k_widen')]
    t0 = ks[0][0]
    qs = sorted({k[2] for k in ks})
    print('--- fit of %d kernels, %.3f ms from first start to last end, queues %s' % (len(ks), (max(k[1] for k in ks) - t0) / 1e6, qs))
    for s, e, q, k in ks:
        if e - s < 4000 and not k.startswith('k_noddi'):
            continue
        print('  %-46s queue/stream %-5s %9.1f -> %9.1f us  (%7.1f)' % (k, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
