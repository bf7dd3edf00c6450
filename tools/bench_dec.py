"""Timing of the two fused expanding blocks alone at the bench shape (BASELINE config 3: 4 frames, 1024^2): L10 (16 outputs, input
128^2 x (32 | 128)) and L11 (8 outputs, input 256^2 x (16 | 64)), and at config 5's (2 frames, 2048^2).  NLT_DEC_GENERIC=1 selects the
generic kernel for an A/B."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nlt_amd import capi as C                                    # noqa: E402
from ab_front import time_it                                     # noqa: E402


def main():
    g = torch.Generator(device='cuda').manual_seed(0)
    U = lambda *s: torch.rand(s, device='cuda', generator=g) - 0.5
    for (n, uv) in ((4, 1024), (2, 2048)):
        for c, div in ((16, 8), (8, 4)):
            h = w = uv // div
            x, skip = U(n, h, w, 2 * c), U(n, h, w, 8 * c)
            w2, b2, w1, b1 = U(2, 2, c, 10 * c), U(c), U(2, 2, c, c), U(c)
            out = torch.empty(n, 2 * h, 2 * w, c, device='cuda')
            t = time_it(lambda: C.dec_block_forward(x, 2 * c, skip, 8 * c, n, h, w, w2, b2, w1, b1, c, 0.3, out))
            flops = 2 * n * h * w * 10 * c * 4 * c + 2 * n * 4 * h * w * 4 * c * c
            moved = 4 * n * h * w * (10 * c + 4 * c)
            print("n %d uv %4d  C %2d (input %4d^2): %.4f ms   %5.1f TFLOP/s  %5.2f TB/s   checksum %.6e"
                  % (n, uv, c, h, t, flops / t / 1e9, moved / t / 1e9, float(out.double().sum())))


if __name__ == '__main__':
    main()
