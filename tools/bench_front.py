"""A/B timing of the fused front kernels at the bench shape (BASELINE config 3: 4 frames, 1024^2, k = 4) with HIP events:
front_kernel<true> (fused.hip), front3 float / uint8-store at 2 and 3 waves per SIMD.    python tools/bench_front.py [k]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlt_amd                                                   # noqa: E402
from nlt_amd import capi as C                                    # noqa: E402
from nlt_amd.models import get_model_class                       # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n, h, w = 4, 1024, 1024
    pm = get_model_class('nlt')(nlt_amd.make_config(depth=256, uvh=h, uvw=w, imh=512, imw=512)).build('cuda')
    blob, blob_l2 = pm.plan._front_weights(torch.device('cuda'))
    g = torch.Generator(device='cuda').manual_seed(0)
    F = 8
    R = lambda *s: torch.randint(0, 256, s, device='cuda', generator=g, dtype=torch.uint8)
    diffuse, rgb, cvis, lvis = R(F, h, w, 3), R(F, h, w, 3), R(F, h, w), R(F, h, w)
    ids = torch.arange(n, device='cuda', dtype=torch.int32)
    nn_ids = torch.randint(0, F, (n, k), device='cuda', generator=g, dtype=torch.int32)
    b = C.assemble_batch(diffuse, rgb, cvis, lvis, ids, nn_ids)
    E = lambda *s: torch.empty(s, device='cuda')
    outs = (E(n, h // 2, w // 2, 32), E(n, h, w, 3), E(n, h // 4, w // 4, 32), E(n, k, h // 4, w // 4, 32))
    fl = (b['base'], b['cvis'], b['lvis'], b['nn_rgb'], b['nn_base'])
    variants = {
        'front2 (fused.hip)': lambda: C.front2_forward(*fl, n, k, h, w, blob, blob_l2, True, 0.3, *outs),
        'front4 f32 wps2': lambda: C.front4_forward(*fl, n, k, h, w, blob, blob_l2, True, 0.3, *outs, 2),
        'front4 f32 wps3': lambda: C.front4_forward(*fl, n, k, h, w, blob, blob_l2, True, 0.3, *outs, 3),
        'front4 u8  wps3': lambda: C.front4_forward_u8(diffuse, rgb, cvis, lvis, ids, nn_ids, n, k, h, w, blob, blob_l2, True, 0.3, *outs, 3),
        'front4 u8  wps2': lambda: C.front4_forward_u8(diffuse, rgb, cvis, lvis, ids, nn_ids, n, k, h, w, blob, blob_l2, True, 0.3, *outs, 2),
        'assemble_batch': lambda: C.assemble_batch(diffuse, rgb, cvis, lvis, ids, nn_ids),
    }
    flops = 2 * n * (h // 2) * (w // 2) * ((32 + 64) * 16 + k * (12 + 64) * 16) + 2 * n * h * w * 24 + 2 * n * (h // 4) * (w // 4) * 32 * (128 + 64 * k)
    for name, fn in variants.items():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        t = sorted(ts)[len(ts) // 2]
        print("%-22s %8.4f ms   %6.1f TFLOP/s (front work)" % (name, t, flops / t / 1e9 if 'front' in name else 0.0))


if __name__ == '__main__':
    main()
