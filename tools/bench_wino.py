"""conv_tile (direct sum, fp32 MFMA) / conv_tile3 (9 bf16 term products) vs conv_wino (Winograd F(2x2,2x2), fp32 MFMA) on the
stride-1 encoder / decoder shapes of BASELINE config 3 (4 frames x 4 observations; query: 4 frames).  python tools/bench_wino.py"""
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
for l, (res, c) in enumerate([(512, 16), (256, 32), (128, 64), (64, 128), (32, 256), (16, 256)], 1):
    if l >= 2:
        cases.append(('L%d.o.s1' % l, c, res, 4, 4))
        cases.append(('L%d.q.s1' % l, c, res, 4, 1))
print("%-10s %5s %5s | %9s %7s | %9s %7s | %9s %7s %5s | %9s %7s %5s | %9s" % ('launch', 'c', 'res', 'direct ms', 'TF', 'x3-9 ms', 'TF', 'wino32 ms', 'TF', 'x', 'wino64 ms', 'TF', 'x', '64+mean'))
for name, c, res, frames, kobs in cases:
    g = torch.Generator(device='cuda').manual_seed(1)
    src = torch.randn((frames * kobs, res, res, c), device='cuda', generator=g)
    wk = torch.randn((2, 2, c, c), device='cuda', generator=g) * (0.5 / (c ** 0.5))
    bias = torch.zeros(c, device='cuda')
    out = torch.empty((frames * kobs, res, res, c), device='cuda')
    mean = torch.empty((frames, res, res, c), device='cuda') if kobs > 1 else None
    flops = 2 * frames * kobs * res * res * 4 * c * c
    t1 = t9 = 1e9
    for tn in (32, 64):
        if c % tn or c % 16:
            continue
        p1 = C.pack_conv_tile_weights(C.CONV_K2S1, wk, c, c, tn)
        p3 = C.pack_conv_tile3_weights(C.CONV_K2S1, wk, c, c, tn)
        t1 = min(t1, timeit(lambda: C.conv_tile_forward(C.CONV_K2S1, src, c, c, frames, kobs, res, res, p1, bias, c, tn, out, c, mean, c)))
        t9 = min(t9, timeit(lambda: C.conv_tile3_forward(C.CONV_K2S1, src, c, c, frames, kobs, res, res, p3, bias, c, tn, out, c, mean, c, nprod=9)))
    pw = C.pack_conv_wino_weights(C.CONV_K2S1, wk, c, c, 32)
    tw32 = timeit(lambda: C.conv_wino_forward(C.CONV_K2S1, src, c, c, frames, kobs, res, res, pw, bias, c, 32, out, c, mean, c))
    tw64 = tm = float('nan')
    if c % 64 == 0:
        pw6 = C.pack_conv_wino_weights(C.CONV_K2S1, wk, c, c, 64)
        tw64 = timeit(lambda: C.conv_wino_forward(C.CONV_K2S1, src, c, c, frames, kobs, res, res, pw6, bias, c, 64, out, c, mean, c))
        tm = tw64
    if C.conv_c32_supported(C.CONV_K2S1, c, c):
        pc = C.pack_conv_tile_weights(C.CONV_K2S1, wk, c, c, 32)
        tc = timeit(lambda: C.conv_c32_forward(C.CONV_K2S1, src, c, c, frames, kobs, res, res, pc, bias, c, out, c, mean, c))
        print("%-10s conv_c32 (LDS-resident weights, frame per stage): %.4f ms  %.1f TF  %.2fx" % (name, 1e3 * tc, flops / tc / 1e12, t1 / tc))
    print("%-10s %5d %5d | %9.4f %7.1f | %9.4f %7.1f | %9.4f %7.1f %5.2f | %9.4f %7.1f %5.2f | %9.4f"
          % (name, c, res, 1e3 * t1, flops / t1 / 1e12, 1e3 * t9, flops / t9 / 1e12, 1e3 * tw32, flops / tw32 / 1e12, t1 / tw32,
             1e3 * tw64, flops / tw64 / 1e12, t1 / tw64, 1e3 * tm))
