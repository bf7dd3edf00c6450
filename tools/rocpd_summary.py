#!/usr/bin/env python
"""Turns a rocprofv3 rocpd database (ROCm 7.2 default output of `--kernel-trace --stats`) into the
per-kernel stats CSV kept under profiles/.   usage: rocpd_summary.py results.db out.csv"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'Percentage'])
    for r in rows:
        w.writerow([r[0], r[1], '%.3f' % r[2], '%.3f' % r[3], '%.3f' % r[4]])
print('wrote %d kernels to %s' % (len(rows), sys.argv[2]))
