"""Weight-gradient / backward-data launches of the config-4 train step (4 frames, 1024^2, k = 1, depth 256), one at a
time, back to back (50 queued calls between two events: launch overhead amortised, the reduce passes pipelined behind the
next call as in the real step):  python tools/bench_bwd.py [--only wgrad|dgrad]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlt_amd                                                   # noqa: E402
from nlt_amd import capi as C                                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--only', default=None)
ap.add_argument('--label', default=None, help='only the ops whose label contains this (PMC passes)')
ap.add_argument('--uv', type=int, default=1024)
ap.add_argument('--frames', type=int, default=4)
ap.add_argument('--reps', type=int, default=50)
args = ap.parse_args()


def timeit(fn, reps=args.reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


n, uv = args.frames, args.uv
ch = [16, 16, 32, 64, 128, 256, 512]                             # channels after level l (level 0 = the 1x1 stem)
ops = []                                                         # (label, mode, cin, cout, input h)
for l in range(1, 7):
    hin = uv >> (l - 1)
    ops.append(('L%d.q.s2' % l, C.CONV_K2S2, 2 * ch[l - 1], ch[l], hin))
    ops.append(('L%d.q.s1' % l, C.CONV_K2S1, ch[l], ch[l], hin // 2))
    ops.append(('L%d.o.s2' % l, C.CONV_K2S2, ch[l - 1], ch[l], hin))
for j in range(6):                                               # expanding block j: level 6 - j -> 5 - j
    l = 6 - j
    cx = 2 * ch[6] if j == 0 else ch[l]                          # x (bottleneck self-concat / previous block) ...
    cs = 0 if j == 0 else 2 * ch[l]                              # ... + skip
    cout = ch[l - 1] if l > 1 else 8
    hin = uv >> l
    ops.append(('L%d.q.s2' % (7 + j), C.DECONV_K2S2, cx + cs, cout, hin))
    ops.append(('L%d.q.s1' % (7 + j), C.DECONV_K2S1, cout, cout if l > 1 else 4, 2 * hin))
ADJ = {C.CONV_K2S2: C.DECONV_K2S2, C.CONV_K2S1: C.DECONV_K2S1, C.DECONV_K2S2: C.CONV_K2S2, C.DECONV_K2S1: C.CONV_K2S1}

print('%-10s %5s %5s %5s  %9s %8s   %s' % ('op', 'cin', 'cout', 'h_in', 'GFLOP', 'us', 'TF/s'))
for label, mode, cin, cout, h in ops:
    if args.label and args.label not in label:
        continue
    tr = mode in (C.DECONV_K2S2, C.DECONV_K2S1)
    oh = h // 2 if mode == C.CONV_K2S2 else (2 * h if mode == C.DECONV_K2S2 else h)
    taps = 1 if mode == C.DECONV_K2S2 else 4
    rows = n * h * h // (4 if mode == C.CONV_K2S2 else 1)
    ncols = cout * (4 if mode == C.DECONV_K2S2 else 1)
    gf = 2.0 * rows * taps * cin * ncols / 1e9
    x = torch.randn(n, h, h, cin, device='cuda')
    dp = torch.randn(n, oh, oh, cout, device='cuda')
    dw = torch.zeros((2, 2, cout, cin) if tr else (2, 2, cin, cout), device='cuda')
    db = torch.zeros(cout, device='cuda')
    if args.only in (None, 'wgrad'):
        narrow = ncols <= 32 and cin * (1 if mode == C.DECONV_K2S2 else 4) <= 128
        fn = C.conv_backward_weights_narrow if narrow else C.conv_backward_weights_tiled
        t = timeit(lambda: fn(mode, x, cin, cin, None, 0, 0, n, h, h, dp, cout, cout, dw, db))
        print('%-10s %5d %5d %5d  %9.2f %8.1f   %6.1f   wgrad %s' % (label, cin, cout, h, gf, t, gf / t * 1e3, 'narrow' if narrow else 'tiled'))
    if args.only in (None, 'dgrad'):
        wk = torch.randn_like(dw) * 0.01
        adj = ADJ[mode]
        # adjoint weights: the packing the engine uses (Conv2D.packed_adjoint) is a permutation of the same bytes; timing only
        wa = torch.randn((2, 2, cin, cout) if adj in (C.DECONV_K2S2, C.DECONV_K2S1) else (2, 2, cout, cin), device='cuda') * 0.01
        packed = C.pack_conv_weights(adj, wa, cout, 0, cin)
        dx = torch.empty(n, h, h, cin, device='cuda')
        zb = torch.zeros(cin, device='cuda')
        res = []
        for tile in (0, 17, 18, 20, 33, 34, 36, 65, 66, 68):
            if tile and ((cin * (4 if adj == C.DECONV_K2S2 else 1) + 15) // 16) % (tile & 15):
                continue
            try:
                t = timeit(lambda: C.conv_forward(adj, dp, cout, cout, None, 0, 0, n, oh, oh, wa, packed, zb, cin, dx, cin, act=False,
                                                  algo=C.ALGO_MFMA, tile_hint=tile, mask_src=x, ldm=cin, alpha=0.3), reps=20)
            except Exception as e:                                # noqa: BLE001
                continue
            res.append((t, tile))
        best = min(res)
        print('%-10s %5d %5d %5d  %9.2f %8.1f   %6.1f   dgrad best tile %d (default %.1f us)' %
              (label, cin, cout, h, gf, best[0], gf / best[0] * 1e3, best[1], res[0][0]))
