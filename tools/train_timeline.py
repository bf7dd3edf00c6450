#!/usr/bin/env python
"""Timeline of the train step from a rocprofv3 kernel trace (CSV):  where the step's wall time goes.

    train_timeline.py trace_dir out.txt [n_last_steps] [delimiter kernel substring]

Steps are delimited by the optimizer kernel (adam_amsgrad_kernel; pass e.g. warp_kernel for forward-only traces).  For the last n steps: wall span, GPU-busy time (union
of the kernel intervals over all streams), sum of kernel durations, number of kernels, and the kernels ranked by total
time; plus the idle gaps (no kernel running on any stream) ranked by the kernel that ENDS the gap."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1], sys.argv[2]
    nlast = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    delim = sys.argv[4] if len(sys.argv) > 4 else 'adam_amsgrad_kernel'
    rows = []
    for path in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        with open(path, newline='') as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id') or r.get('Queue_Id')))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if delim in r[2]]
    if len(ends) < nlast + 1:
        raise SystemExit('only %d %s launches in the trace' % (len(ends), delim))
    lines = []
    per_kernel = defaultdict(lambda: [0, 0.0])
    gap_after = defaultdict(lambda: [0, 0.0])
    spans, busys, sums, counts = [], [], [], []
    for s in range(len(ends) - nlast, len(ends)):
        a, b = ends[s - 1] + 1, ends[s] + 1
        ks = rows[a:b]
        t0, t1 = rows[ends[s - 1]][1], ks[-1][1]
        busy, cur_end, tot = 0, t0, 0
        for st, en, name, _ in ks:
            tot += en - st
            short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            per_kernel[short][0] += 1
            per_kernel[short][1] += (en - st) / 1e3
            if st > cur_end:
                gap_after[short][0] += 1
                gap_after[short][1] += (st - cur_end) / 1e3
            lo = max(st, cur_end)
            if en > lo:
                busy += en - lo
                cur_end = en
        spans.append((t1 - t0) / 1e3); busys.append(busy / 1e3); sums.append(tot / 1e3); counts.append(len(ks))
    n = float(nlast)
    lines.append('steps analysed: %d   (per step averages, microseconds)' % nlast)
    lines.append('wall span %.1f   gpu busy (union over streams) %.1f   idle %.1f   sum of kernel durations %.1f   kernels %.1f'
                 % (sum(spans) / n, sum(busys) / n, (sum(spans) - sum(busys)) / n, sum(sums) / n, sum(counts) / n))
    lines.append('')
    lines.append('%-70s %8s %10s %9s' % ('kernel', 'calls', 'us/step', 'avg us'))
    for name, (c, t) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        lines.append('%-70s %8.1f %10.1f %9.1f' % (name[:70], c / n, t / n, t / c))
    # per hardware queue (= stream) of the LAST analysed step: busy time, first start / last end relative to the step start
    s = len(ends) - 1
    ks = rows[ends[s - 1] + 1:ends[s] + 1]
    t0 = rows[ends[s - 1]][1]
    per_q = defaultdict(lambda: [0.0, None, 0.0, 0])
    for st, en, name, q in ks:
        r = per_q[q]
        r[0] += (en - st) / 1e3
        r[1] = (st - t0) / 1e3 if r[1] is None else r[1]
        r[2] = (en - t0) / 1e3
        r[3] += 1
    lines.append('')
    lines.append('streams of the last step: queue, kernels, sum of durations us, first start us, last end us')
    for q, r in sorted(per_q.items(), key=lambda kv: -kv[1][0]):
        lines.append('  queue %-6s %5d %10.1f %10.1f %10.1f' % (q, r[3], r[0], r[1], r[2]))
    lines.append('')
    lines.append('kernels of the last step longer than 40 us: start us, duration us, queue, name')
    for st, en, name, q in ks:
        if en - st > 40000:
            lines.append('  %8.1f %7.1f  q%-4s %s' % ((st - t0) / 1e3, (en - st) / 1e3, q,
                                                   name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]))
    lines.append('')
    lines.append('loss window of the last step (every kernel from the resampler to the first backward launch): start us, duration us, queue, name')
    on = False
    for st, en, name, q in ks:
        short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:70]
        if short.startswith('warp_kernel'):
            on = True
        if on:
            lines.append('  %8.1f %7.1f  q%-4s %s' % ((st - t0) / 1e3, (en - st) / 1e3, q, short))
        if on and 'back_bwd_kernel' in short:
            break
    lines.append('')
    lines.append('idle gaps (no kernel on any stream), by the kernel that ends the gap')
    for name, (c, t) in sorted(gap_after.items(), key=lambda kv: -kv[1][1])[:25]:
        lines.append('%-70s %8.1f %10.1f %9.1f' % (name[:70], c / n, t / n, t / c))
    with open(out, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:8]))
    i = lines.index('streams of the last step: queue, kernels, sum of durations us, first start us, last end us')
    print('\n'.join(lines[i:i + 6]))
    i = [n for n, l in enumerate(lines) if l.startswith('loss window')][0]
    print('\n'.join(lines[i:i + 30]))


if __name__ == '__main__':
    main()
