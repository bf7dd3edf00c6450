#!/usr/bin/env python
"""Every kernel of the LAST forward step of a rocprofv3 kernel trace: start, duration, queue, name (steps delimited by warp_kernel).
    step_kernels.py trace_dir [delimiter]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
delim = sys.argv[2] if len(sys.argv) > 2 else 'warp_kernel'
rows = []
for path in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id') or r.get('Stream_Id')))
rows.sort()
ends = [i for i, r in enumerate(rows) if delim in r[2]]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = rows[ends[-2]][1]
queues = {}
for st, en, name, q in rows[a:b]:
    short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    print("%8.1f %7.1f  q%d  %s" % ((st - t0) / 1e3, (en - st) / 1e3, queues.setdefault(q, len(queues)), short[:70]))
print("step wall %.1f us" % ((rows[b - 1][1] - t0) / 1e3))
