#!/usr/bin/env python
"""BASELINE config 5 stress point: bf16 1x1 channel mix (64 -> 64) over a 2048^2 UV grid on one MI355X.
Prints one JSON line: Mtexels/s, HBM GB/s by algorithmic bytes (cin + cout bf16 per texel) and the fraction of
the 8 TB/s peak.   python tools/bench_chmix.py [--uv 2048 --frames 1 --c 64 --reps 50]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nlt_amd
from nlt_amd import capi as C

ap = argparse.ArgumentParser()
ap.add_argument('--uv', type=int, default=2048); ap.add_argument('--frames', type=int, default=1)
ap.add_argument('--c', type=int, default=64); ap.add_argument('--reps', type=int, default=50)
a = ap.parse_args()
dev = torch.device('cuda', 0)
x = (torch.randn(a.frames, a.uv, a.uv, a.c, device=dev) * 0.7).to(torch.bfloat16)
w = torch.randn(1, 1, a.c, a.c, device=dev) * a.c ** -0.5
b = torch.randn(a.c, device=dev) * 0.1
pk = C.chmix_bf16_pack(w)
for _ in range(5):
    y = C.chmix_bf16_forward(x, pk, b, a.c)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(a.reps):
    y = C.chmix_bf16_forward(x, pk, b, a.c)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
texels = a.frames * a.uv * a.uv
nbytes = texels * 2 * 2 * a.c
print(json.dumps({"metric": "bf16 1x1 channel mix %d->%d at %d^2 UV" % (a.c, a.c, a.uv), "value": round(texels / ms / 1e3, 1),
                  "unit": "Mtexels/s", "ms": round(ms, 4), "dtype": "bf16 (fp32 accumulate)",
                  "roofline": {"bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(nbytes / ms / 1e6 / 8000.0, 4), "algorithmic_bytes_per_launch": nbytes},
                  "mfma_tflops": round(2.0 * texels * a.c * a.c / ms / 1e9, 2)}))
