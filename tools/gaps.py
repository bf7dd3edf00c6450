#!/usr/bin/env python
"""Gaps between consecutive kernels of each queue in a step_kernels.py listing:  gaps.py listing.txt [min_gap_us]"""
import sys
rows = []
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) >= 4 and p[2].startswith('q') and p[2][1:].isdigit():
        rows.append((float(p[0]), float(p[1]), p[2], ' '.join(p[3:])))
mg = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
for q in sorted({r[2] for r in rows}):
    qs = [r for r in rows if r[2] == q]
    tot = n = 0
    for a, b in zip(qs, qs[1:]):
        gap = b[0] - (a[0] + a[1])
        if gap > mg:
            n += 1
            tot += gap
    busy = sum(r[1] for r in qs)
    print("%s: %d kernels, busy %.1f us, %d gaps > %.1f us totalling %.1f us, span %.1f..%.1f" % (q, len(qs), busy, n, mg, tot, qs[0][0], qs[-1][0] + qs[-1][1]))
