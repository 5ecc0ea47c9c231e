#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace like `--stats` does: calls, total / avg /
min / max duration per kernel, plus registers / LDS / scratch.  usage: rocpd_summary.py X.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select name, count(*), sum(duration), avg(duration), min(duration), max(duration),
                     max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size),
                     max(grid_x), max(workgroup_x)
                     from kernels group by name order by sum(duration) desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
print('%-78s %6s %12s %12s %12s %12s %6s %5s %5s %5s %7s %7s %9s %4s' % (
    'kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'pct', 'vgpr', 'agpr', 'sgpr', 'lds', 'scratch', 'grid', 'wg'))
for r in rows:
    print('%-78s %6d %12d %12d %12d %12d %6.2f %5d %5d %5d %7d %7d %9d %4d' % (
        r[0][:78], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
