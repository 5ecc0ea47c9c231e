#!/usr/bin/env python3
"""Timeline of the LAST host-buffer call in a rocprofv3 csv trace (kernel_trace + memory_copy_trace)."""
import csv
import glob
import sys

root = sys.argv[1]
ev = []
for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:50]))
for f in glob.glob(root + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C %s %s' % (r.get('Direction', ''), r.get('Size', r.get('Bytes', '')))))
ev.sort()
# calls are separated by the big D2H of the estimates: take events after the third-last gap > 5 ms ... simpler: last 45 ms
end = ev[-1][1]
# find start of last call: last idle gap > 1 ms going backwards beyond 30 ms
cut = 0
for i in range(len(ev) - 1, 0, -1):
    if end - ev[i][0] > 30e6 and ev[i][0] - max(e[1] for e in ev[:i]) > 0.3e6:
        cut = i
        break
sel = ev[cut:]
t0 = sel[0][0]
print('events in last call: %d, span %.2f ms' % (len(sel), (end - t0) / 1e6))
busy_k = sum(e[1] - e[0] for e in sel if e[2][0] == 'K')
busy_c = sum(e[1] - e[0] for e in sel if e[2][0] == 'C')
print('kernel time %.2f ms, copy time %.2f ms' % (busy_k / 1e6, busy_c / 1e6))
for s, e, nme in sel:
    if e - s > 20000 or nme[0] == 'C':
        print('%9.3f %9.3f  %s' % ((s - t0) / 1e6, (e - s) / 1e6, nme))
