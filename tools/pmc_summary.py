#!/usr/bin/env python
"""Summarises rocprofv3 --pmc passes (counter_collection CSVs) into per-kernel averages and HBM bytes.

    pmc_summary.py out.json fetch_dir write_dir

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports
half of the bytes of a wide coalesced streaming read -> doubled here (`fetch_corrected`); WRITE_SIZE is used
as reported (uncalibrated, said so in the JSON)."""
import csv
import glob
import json
import os
import sys


def collect(d):
    acc = {}
    for path in glob.glob(os.path.join(d, '**', '*counter_collection*.csv'), recursive=True):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                name = row.get('Kernel_Name') or row.get('Kernel Name') or ''
                cname = row.get('Counter_Name') or ''
                try:
                    val = float(row.get('Counter_Value') or 'nan')
                except ValueError:
                    continue
                key = (name, cname)
                disp = row.get('Dispatch_Id') or row.get('Correlation_Id') or str(len(acc))
                acc.setdefault(key, {}).setdefault(disp, 0.0)
                acc[key][disp] += val                      # one row per (dispatch, counter[, dimension instance])
    return acc


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    per = {}
    for d in dirs:
        for (name, cname), disp in collect(d).items():
            vals = list(disp.values())
            per.setdefault(name, {})[cname] = {'calls': len(vals), 'avg': sum(vals) / len(vals), 'max': max(vals)}
    kernels = {}
    for name, c in per.items():
        rec = {'counters': c}
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            # steady-state launches: take the MAX-call-count kernels' average over all calls
            rec['fetch_bytes_reported'] = c['FETCH_SIZE']['avg'] * 1024
            rec['fetch_bytes_corrected'] = 2 * c['FETCH_SIZE']['avg'] * 1024
            rec['write_bytes_reported'] = c['WRITE_SIZE']['avg'] * 1024
            rec['hbm_bytes'] = rec['fetch_bytes_corrected'] + rec['write_bytes_reported']
        kernels[name[:200]] = rec
    # whole forward step: every kernel of the profiled command, per forward step (= launches of the front kernel; the
    # profiled command loads its tile choices from a cache, so no plan-time trial launches are in the trace)
    steps = max([rec['counters']['FETCH_SIZE']['calls'] for name, rec in kernels.items()
                 if ('front4_kernel' in name or 'front_kernel' in name) and 'FETCH_SIZE' in rec['counters']] or [0])
    total = sum(rec['hbm_bytes'] * rec['counters']['FETCH_SIZE']['calls'] for rec in kernels.values() if 'hbm_bytes' in rec)
    with open(out_path, 'w') as f:
        json.dump({'forward_steps_profiled': steps, 'hbm_bytes_per_forward_step': total / steps if steps else None,
                   'note': 'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); '
                           'WRITE_SIZE as reported (uncalibrated); KiB -> bytes', 'kernels': kernels}, f, indent=1)
    for name, rec in sorted(kernels.items(), key=lambda kv: -kv[1].get('hbm_bytes', 0))[:12]:
        print('%-60s %s' % (name[:60], {k: round(v / 1e6, 2) for k, v in rec.items() if k.endswith('bytes') or k.endswith('corrected') or k.endswith('reported')}))


if __name__ == '__main__':
    main()
