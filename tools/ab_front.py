"""Timing of the fused front kernel alone (BASELINE config 3 shape: 4 frames, 1024^2) for k = 1, 2, 4 observations, float and
uint8-store inputs, and at config 5's shape (2 frames, 2048^2, k = 1).  The k sweep separates the per-observation cost from
everything else (prologue, query path).  NLT_HIP_LIB=<other build> python tools/ab_front.py for an A/B on one box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlt_amd                                                   # noqa: E402
from nlt_amd import capi as C                                    # noqa: E402
from nlt_amd.models import get_model_class                       # noqa: E402


def time_it(fn, reps=5, inner=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]


def main():
    print("lib:", C.LIB_PATH)
    res = {}
    for (n, h, w, ks) in ((4, 1024, 1024, (1, 2, 4)), (2, 2048, 2048, (1,))):
        pm = get_model_class('nlt')(nlt_amd.make_config(depth=256, uvh=h, uvw=w, imh=512, imw=512)).build('cuda')
        blob, blob_l2 = pm.plan._front_weights(torch.device('cuda'))
        g = torch.Generator(device='cuda').manual_seed(0)
        F = 8
        R = lambda *s: torch.randint(0, 256, s, device='cuda', generator=g, dtype=torch.uint8)
        diffuse, rgb, cvis, lvis = R(F, h, w, 3), R(F, h, w, 3), R(F, h, w), R(F, h, w)
        ids = torch.arange(n, device='cuda', dtype=torch.int32)
        for k in ks:
            nn_ids = torch.randint(0, F, (n, k), device='cuda', generator=g, dtype=torch.int32)
            b = C.assemble_batch(diffuse, rgb, cvis, lvis, ids, nn_ids)
            E = lambda *s: torch.empty(s, device='cuda')
            outs = (E(n, h // 2, w // 2, 32), E(n, h, w, 3), E(n, h // 4, w // 4, 32), E(n, k, h // 4, w // 4, 32))
            fl = (b['base'], b['cvis'], b['lvis'], b['nn_rgb'], b['nn_base'])
            t32 = time_it(lambda: C.front4_forward(*fl, n, k, h, w, blob, blob_l2, True, 0.3, *outs, 2))
            cs = [float(o.double().sum()) for o in outs]
            t8 = time_it(lambda: C.front4_forward_u8(diffuse, rgb, cvis, lvis, ids, nn_ids, n, k, h, w, blob, blob_l2, True, 0.3, *outs, 2))
            print("n %d  %4d^2  k %d:  f32 %.4f ms   u8 %.4f ms   checksums %s" % (n, h, k, t32, t8, ' '.join('%.6e' % c for c in cs)))
            res[(h, k)] = t32
    if (1024, 1) in res and (1024, 4) in res:
        per = (res[(1024, 4)] - res[(1024, 1)]) / 3
        print("1024^2: per observation %.4f ms, everything else %.4f ms" % (per, res[(1024, 1)] - per))


if __name__ == '__main__':
    main()
