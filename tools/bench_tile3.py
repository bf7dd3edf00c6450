"""conv_tile (native fp32 MFMA) vs conv_tile3 (three-term bf16 split, 6 / 9 products) on the encoder shapes of BASELINE config 3
(4 frames x 4 observations = 16 observation frames per level; query: 4 frames).  python tools/bench_tile3.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nlt_amd import capi as C                                    # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


cases = []
for l, (res, cp, c) in enumerate([(1024, 16, 16), (512, 16, 32), (256, 32, 64), (128, 64, 128), (64, 128, 256), (32, 256, 256)], 1):
    # res = resolution of level l - 1's maps (the input of level l's stride-2 conv) at 1024^2 UV
    if l >= 2:
        cases.append(('L%d.o.s2' % l, C.CONV_K2S2, cp, c, res, 4, 4))
        cases.append(('L%d.o.s1' % l, C.CONV_K2S1, c, c, res // 2, 4, 4))
        cases.append(('L%d.q.s2' % l, C.CONV_K2S2, 2 * cp, c, res, 4, 1))
        cases.append(('L%d.q.s1' % l, C.CONV_K2S1, c, c, res // 2, 4, 1))
print("%-10s %5s %5s %5s | %9s %7s | %9s %7s %5s | %9s %7s %5s | %9s" % ('launch', 'cin', 'cout', 'res', 'fp32 ms', 'TF', 'x3-6 ms', 'TF', 'x', 'x3-9 ms', 'TF', 'x', 'x3-1 ms'))
for name, mode, cin, cout, res, frames, kobs in cases:
    if cin % 16:
        continue
    g = torch.Generator(device='cuda').manual_seed(1)
    src = torch.randn((frames * kobs, res, res, cin), device='cuda', generator=g)
    wk = torch.randn((2, 2, cin, cout), device='cuda', generator=g) * (0.5 / (cin ** 0.5))
    bias = torch.zeros(cout, device='cuda')
    oh = res // 2 if mode == C.CONV_K2S2 else res
    out = torch.empty((frames * kobs, oh, oh, cout), device='cuda')
    mean = torch.empty((frames, oh, oh, cout), device='cuda') if kobs > 1 else None
    flops = 2 * frames * kobs * oh * oh * 4 * cin * cout
    best = {}
    for tn in (32, 64):
        if cout % tn:
            continue
        p1 = C.pack_conv_tile_weights(mode, wk, cin, cout, tn)
        p3 = C.pack_conv_tile3_weights(mode, wk, cin, cout, tn)
        t1 = timeit(lambda: C.conv_tile_forward(mode, src, cin, cin, frames, kobs, res, res, p1, bias, cout, tn, out, cout, mean, cout))
        t6 = timeit(lambda: C.conv_tile3_forward(mode, src, cin, cin, frames, kobs, res, res, p3, bias, cout, tn, out, cout, mean, cout, nprod=6))
        t9 = timeit(lambda: C.conv_tile3_forward(mode, src, cin, cin, frames, kobs, res, res, p3, bias, cout, tn, out, cout, mean, cout, nprod=9))
        th = timeit(lambda: C.conv_tile3_forward(mode, src, cin, cin, frames, kobs, res, res, p3, bias, cout, tn, out, cout, mean, cout, nprod=1))
        for k_, t in (('1', t1), ('6', t6), ('9', t9), ('h', th)):
            best[k_] = min(best.get(k_, 1e9), t)
    t1, t6, t9 = best['1'], best['6'], best['9']
    print("%-10s %5d %5d %5d | %9.4f %7.1f | %9.4f %7.1f %5.2f | %9.4f %7.1f %5.2f | %9.4f"
          % (name, cin, cout, res, 1e3 * t1, flops / t1 / 1e12, 1e3 * t6, flops / t6 / 1e12, t1 / t6, 1e3 * t9, flops / t9 / 1e12, t1 / t9, 1e3 * best['h']))
