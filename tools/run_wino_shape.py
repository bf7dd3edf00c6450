"""One stride-1 k2 conv shape on each kernel family, a few launches each (driver of tools/pmc_wino.sh).
python tools/run_wino_shape.py C RES FRAMES KOBS"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nlt_amd import capi as C                                    # noqa: E402

c, res, frames, kobs = (int(a) for a in sys.argv[1:5])
g = torch.Generator(device='cuda').manual_seed(1)
src = torch.randn((frames * kobs, res, res, c), device='cuda', generator=g)
wk = torch.randn((2, 2, c, c), device='cuda', generator=g) * (0.5 / (c ** 0.5))
bias = torch.zeros(c, device='cuda')
out = torch.empty((frames * kobs, res, res, c), device='cuda')
mean = torch.empty((frames, res, res, c), device='cuda') if kobs > 1 else None
for tn in (32, 64):
    if c % tn:
        continue
    p1 = C.pack_conv_tile_weights(C.CONV_K2S1, wk, c, c, tn)
    p3 = C.pack_conv_tile3_weights(C.CONV_K2S1, wk, c, c, tn)
    pw = C.pack_conv_wino_weights(C.CONV_K2S1, wk, c, c, tn)
    for _ in range(4):
        C.conv_tile_forward(C.CONV_K2S1, src, c, c, frames, kobs, res, res, p1, bias, c, tn, out, c, mean, c)
        C.conv_tile3_forward(C.CONV_K2S1, src, c, c, frames, kobs, res, res, p3, bias, c, tn, out, c, mean, c, nprod=9)
        C.conv_wino_forward(C.CONV_K2S1, src, c, c, frames, kobs, res, res, pw, bias, c, tn, out, c, mean, c)
        C.conv_wino_forward(C.CONV_K2S1, src, c, c, frames * kobs, 1, res, res, pw, bias, c, tn, out, c, None, 0)
torch.cuda.synchronize()
