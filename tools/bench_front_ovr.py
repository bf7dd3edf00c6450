"""nlt_front_ovr_forward alone: python tools/bench_front_ovr.py [uv] [frames]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nlt_amd import _capi as C                                    # noqa: E402

uv = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.rand(s, device=dev, generator=g) - 0.5
h = w = uv
base, cvis, lvis = R(n, h, w, 3), R(n, h, w, 1), R(n, h, w, 1)
z = lambda *s: torch.zeros(s, device=dev)
blob = C.front_pack_weights(R(1, 1, 5, 16), R(16), z(1, 1, 3, 16), z(16), R(2, 2, 32, 16), R(16), R(2, 2, 16, 16), R(16),
                            z(2, 2, 16, 16), z(16), z(2, 2, 16, 16), z(16), R(1, 1, 36, 3), R(3))
blob2 = C.front_pack_l2_weights(R(2, 2, 32, 32), R(32), z(2, 2, 16, 32), z(32))
p1, s0, p2 = R(1, h // 2, w // 2, 16), R(1, h, w, 4), R(1, h // 4, w // 4, 32)
fm1 = torch.empty((n, h // 2, w // 2, 32), device=dev)
skip3 = torch.empty((n, h, w, 3), device=dev)
qtmp2 = torch.empty((n, h // 4, w // 4, 32), device=dev)
run = lambda: C.front_ovr_forward(base, cvis, lvis, n, h, w, blob, blob2, p1, s0, p2, True, 0.3, fm1, 32, skip3, qtmp2)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
moved = 4 * n * h * w * (5 + 3) + 4 * n * (h // 2) * (w // 2) * 16 + 4 * n * (h // 4) * (w // 4) * 32 + 4 * (p1.numel() + s0.numel() + p2.numel())
flops = 2 * n * (h // 2) * (w // 2) * (20 + 64) * 16 + 2 * n * h * w * 15 + 2 * n * (h // 4) * (w // 4) * 64 * 32
print("front_ovr %dx%d^2: %.4f ms  %.1f GB/s of its own traffic  %.1f TFLOP/s  (NLT_FRONT_OVR=%s)"
      % (n, uv, ms, moved / ms / 1e6, flops / ms / 1e9, os.environ.get('NLT_FRONT_OVR', '')))
