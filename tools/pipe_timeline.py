#!/usr/bin/env python
"""Overlap summary of a rocprofv3 kernel trace of `bench.py --headline-only --pipelined`:  pipe_timeline.py trace_dir out.txt
The trace holds the headline's steps (one batch at a time) and then the pipelined legs.  For the window of the last
`nwarp` resampler launches of each phase: wall time per batch, mean number of kernels running at once, and for the front
kernel, the HBM-bound tail kernels and the middle: how much of their run time another kernel (of another batch) is running too,
and their average duration."""
import csv
import glob
import os
import sys


def load(d):
    rows = []
    for path in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        with open(path, newline='') as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id') or r.get('Stream_Id')))
    rows.sort()
    return rows


def window(rows, lo, hi):
    t0, t1 = rows[lo][0], max(r[1] for r in rows[lo:hi])
    ks = rows[lo:hi]
    ev = sorted([(s, 1) for s, e, _, _ in ks] + [(e, -1) for s, e, _, _ in ks])
    busy = conc = 0.0
    depth, last = 0, t0
    for t, dlt in ev:
        if depth > 0:
            busy += t - last
            conc += depth * (t - last)
        depth += dlt
        last = t
    groups = {'front4_kernel': 'front', 'back_kernel': 'tail', 'warp_kernel': 'tail', 'dec_block_kernel': 'tail'}
    stat = {'front': [0, 0.0, 0.0], 'tail': [0, 0.0, 0.0], 'middle': [0, 0.0, 0.0]}
    for i, (s, e, name, q) in enumerate(ks):
        g = next((v for k, v in groups.items() if k in name), 'middle')
        ov = 0.0                                              # time of [s, e) during which another kernel runs
        segs = sorted((max(s, s2), min(e, e2)) for j, (s2, e2, _, _) in enumerate(ks) if j != i and s2 < e and e2 > s)
        cur = s
        for a, b in segs:
            a = max(a, cur)
            if b > a:
                ov += b - a
                cur = b
        st = stat[g]
        st[0] += 1; st[1] += e - s; st[2] += ov
    return (t1 - t0) / 1e3, busy / 1e3, conc / max(busy, 1), stat, len({r[3] for r in ks})


def main():
    rows = load(sys.argv[1])
    warps = [i for i, r in enumerate(rows) if 'warp_kernel' in r[2]]
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    out = []
    # the trace ends with the widest pipelined leg; the headline's own steps come first
    for label, hi_idx in (('one batch at a time (headline steps)', nsteps + 20), ('last pipelined leg', len(warps) - 1)):
        lo_idx = hi_idx - nsteps
        lo, hi = warps[lo_idx] + 1, warps[hi_idx] + 1
        span, busy, conc, stat, nq = window(rows, lo, hi)
        out.append('%s: %d batches, %.1f us per batch, GPU busy %.1f %%, %.2f kernels running on average while busy, %d hardware queues'
                   % (label, nsteps, span / nsteps, 100 * busy / span, conc, nq))
        for g in ('front', 'middle', 'tail'):
            c, t, ov = stat[g]
            if c:
                out.append('    %-6s %5.1f launches / batch, avg %7.1f us, %5.1f %% of its run time beside another kernel'
                           % (g, c / nsteps, t / c / 1e3, 100 * ov / t))
    open(sys.argv[2], 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out))


if __name__ == '__main__':
    main()
