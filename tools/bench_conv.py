#!/usr/bin/env python
"""Micro-benchmark of ONE encoder conv launch: LDS-tiled kernel (csrc/conv_tile.hip; observations folded or not)
vs the register-tiled MFMA kernel (csrc/conv_mfma.hip, every wave tile).  Run on the GPU box:
    python tools/bench_conv.py --mode s1 --cin 64 --cout 64 --h 128 --w 128 --frames 4 --kobs 4
Level shapes of BASELINE config 3 (4 frames, k = 4): --preset L3.o.s1 etc."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nlt_amd
from nlt_amd import capi as C

PRESETS = {}
cl = [16, 16, 32, 64, 128, 256, 256]
for l in range(2, 7):
    r = 1024 >> l
    PRESETS['L%d.o.s2' % l] = ('s2', cl[l - 1], cl[l], 2 * r, 2 * r, 4, 4)
    PRESETS['L%d.o.s1' % l] = ('s1', cl[l], cl[l], r, r, 4, 4)
    PRESETS['L%d.q.s2' % l] = ('s2', 2 * cl[l - 1], cl[l], 2 * r, 2 * r, 4, 1)
    PRESETS['L%d.q.s1' % l] = ('s1', cl[l], cl[l], r, r, 4, 1)

ap = argparse.ArgumentParser()
ap.add_argument('--preset', default=None)
ap.add_argument('--mode', default='s1'); ap.add_argument('--cin', type=int, default=64); ap.add_argument('--cout', type=int, default=64)
ap.add_argument('--h', type=int, default=128); ap.add_argument('--w', type=int, default=128)
ap.add_argument('--frames', type=int, default=4); ap.add_argument('--kobs', type=int, default=4)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--no-reg', action='store_true', help='LDS-tiled variants only')
ap.add_argument('--only', default=None, help='run only this variant (for rocprofv3 --pmc passes), e.g. lds64f')
a = ap.parse_args()
names = a.preset.split(',') if a.preset else [None]
for name in names:
    if name:
        a.mode, a.cin, a.cout, a.h, a.w, a.frames, a.kobs = PRESETS[name]
    mode = C.CONV_K2S2 if a.mode == 's2' else C.CONV_K2S1
    dev = torch.device('cuda', 0)
    nf = a.frames * a.kobs
    oh, ow = (a.h // 2, a.w // 2) if a.mode == 's2' else (a.h, a.w)
    x = torch.randn(nf, a.h, a.w, a.cin, device=dev)
    wk = torch.randn(2, 2, a.cin, a.cout, device=dev) * (4 * a.cin) ** -0.5
    bias = torch.randn(a.cout, device=dev) * 0.1
    out = torch.empty(nf, oh, ow, a.cout, device=dev)
    mean = torch.empty(a.frames, oh, ow, a.cout, device=dev)
    flops = 2.0 * nf * oh * ow * 4 * a.cin * a.cout
    nbytes = 4.0 * (x.numel() + out.numel())

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(a.reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    res = {}
    for tn in (32, 64, 128):
        if a.cout % tn or (tn == 128 and a.mode != 's2'):
            continue
        try:
            pk = C.pack_conv_tile_weights(mode, wk, a.cin, a.cout, tn)
        except C.NLTError:
            continue
        variants = {'lds%df' % tn: lambda pk=pk, tn=tn: C.conv_tile_forward(mode, x, a.cin, a.cin, a.frames, a.kobs, a.h, a.w, pk, bias, a.cout, tn, out, a.cout, mean if a.kobs > 1 else None, a.cout),
                    'lds%du' % tn: lambda pk=pk, tn=tn: C.conv_tile_forward(mode, x, a.cin, a.cin, nf, 1, a.h, a.w, pk, bias, a.cout, tn, out, a.cout, None, 0)}
        for vn, fn in variants.items():
            if a.only is None or a.only == vn:
                res[vn] = timeit(fn)
    if not a.no_reg and (a.only is None or a.only.startswith('reg')):
        pw = C.pack_conv_weights(mode, wk, a.cin, 0, a.cout)
        ntiles = (a.cout + 15) // 16
        for r in (1, 2, 4):
            for c in (1, 2, 4):
                if ntiles % c or (a.only and a.only != 'reg%dx%d' % (r, c)):
                    continue
                res['reg%dx%d' % (r, c)] = timeit(lambda r=r, c=c: C.conv_forward(mode, x, a.cin, a.cin, None, 0, 0, nf, a.h, a.w, wk, pw, bias, a.cout, out, a.cout, tile_hint=16 * r + c))
    best = min(res.values())
    print('%-9s %s cin %d cout %d %dx%d frames %dx%d  %.2f GFLOP %.0f MB' % (name or '', a.mode, a.cin, a.cout, a.h, a.w, a.frames, a.kobs, flops / 1e9, nbytes / 1e6))
    for vn, t in sorted(res.items(), key=lambda kv: kv[1]):
        print('    %-8s %8.4f ms %7.1f TFLOP/s %7.1f GB/s%s' % (vn, t, flops / t / 1e9, nbytes / t / 1e6, '  <-' if t == best else ''))
