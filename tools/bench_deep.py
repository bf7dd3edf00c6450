"""Split-K sweep of the deep small-M convs (depth 1024, 256^2, 4 frames):  python tools/bench_deep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlt_amd                                                   # noqa: E402
from nlt_amd import capi as C                                    # noqa: E402


def timeit(fn, reps=20):
    """GPU-side time per launch: `reps` launches captured in one hipGraph (eager back-to-back launches through ctypes cost ~7 us of
    host time each and would measure that instead)."""
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps)


shapes = [('L8.q.s2', C.CONV_K2S2, 4, 2, 2, 2048, 1024), ('L8.q.s1', C.CONV_K2S1, 4, 1, 1, 1024, 1024),
          ('L7.q.s2', C.CONV_K2S2, 4, 4, 4, 1024, 1024), ('L6.q.s1', C.CONV_K2S1, 4, 4, 4, 512, 512),
          ('L9.q.s2', C.DECONV_K2S2, 4, 1, 1, 4096, 512)]
for name, mode, n, h, w, cin, cout in shapes:
    tr = mode in (C.DECONV_K2S2, C.DECONV_K2S1)
    wk = torch.randn((2, 2, cout, cin) if tr else (2, 2, cin, cout), device='cuda') * 0.01
    x = torch.randn(n, h, w, cin, device='cuda')
    bias = torch.zeros(cout, device='cuda')
    packed = C.pack_conv_weights(mode, wk, cin, 0, cout)
    oh, ow = (h // 2, w // 2) if mode == C.CONV_K2S2 else ((2 * h, 2 * w) if mode == C.DECONV_K2S2 else (h, w))
    out = torch.empty(n, oh, ow, cout, device='cuda')
    mb = wk.numel() * 4 / 1e6
    res = []
    for tile in (17, 18, 20, 34):
        t = timeit(lambda: C.conv_forward(mode, x, cin, cin, None, 0, 0, n, h, w, wk, packed, bias, cout, out, cout, tile_hint=tile))
        res.append(('t%d' % tile, t))
        for ks in (4, 16, 32, 64, 128):
            t = timeit(lambda: C.conv_forward_splitk(mode, ks, x, cin, cin, None, 0, 0, n, h, w, packed, bias, cout, out, cout, tile_hint=tile))
            res.append(('t%d/k%d' % (tile, ks), t))
    best = min(res, key=lambda r: r[1])
    print("%s weights %.1f MB: best %s %.1f us (%.2f TB/s); " % (name, mb, best[0], 1e3 * best[1], mb / best[1] / 1e3) +
          ' '.join('%s=%.0f' % (k, 1e3 * v) for k, v in res))
