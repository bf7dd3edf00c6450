"""Ablation timings of conv_wino_kernel (64-channel form; NLT_WINO_ABL is read once per process: run one process per value).
for a in 0 1 2 3 4 5; do NLT_WINO_ABL=$a python tools/abl_wino.py; done"""
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


row = []
for c, res, nf in ((64, 128, 16), (128, 64, 16), (256, 32, 16), (64, 128, 4)):
    g = torch.Generator(device='cuda').manual_seed(1)
    src = torch.randn((nf, res, res, c), device='cuda', generator=g)
    wk = torch.randn((2, 2, c, c), device='cuda', generator=g) * (0.5 / (c ** 0.5))
    bias = torch.zeros(c, device='cuda')
    out = torch.empty((nf, res, res, c), device='cuda')
    pw = C.pack_conv_wino_weights(C.CONV_K2S1, wk, c, c, 64)
    t = timeit(lambda: C.conv_wino_forward(C.CONV_K2S1, src, c, c, nf, 1, res, res, pw, bias, c, 64, out, c, None, 0))
    row.append("c%d/%d^2x%d %.4f ms" % (c, res, nf, 1e3 * t))
print("ABL=%s  " % os.environ.get('NLT_WINO_ABL', '0') + "   ".join(row))
