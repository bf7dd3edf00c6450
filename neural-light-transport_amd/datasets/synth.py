"""Seeded synthetic capture in the resident uint8 store layout (`datasets/nlt.py:Dataset`): stand-in for a decoded
dragon capture when none is on disk (bench.py, smoke tests).  Value ranges mimic what `data_gen` writes: uint8 texel
buffers, fp16 `uv2cam` maps with 30 % background pixels at (0,0) (data_gen/render.py:155, data_gen/util.py:67-70).

The `uv2cam` map of a real capture is a rendered UV-coordinate pass: piecewise smooth (one affine-ish patch per atlas
chart seen by the camera, seams between charts) with the background as one connected region.  `warp='charts'` (the
default) imitates that: the camera image is cut into chart x chart blocks, every block maps to its own slot of the unit
UV square (a random permutation of the slots per frame, randomly flipped / transposed), foreground = a centred disc of
area fg_frac.  `warp='random'` is the adversarial case (independent uniform coordinates per pixel, per-pixel random
background): every resampler tap and every backward scatter-add lands on its own cache line."""
import math

import torch


def chart_warp(n_frames, cam, g, device, fg_frac=0.7, chart=64):
    chart = min(chart, cam)
    nb = cam // chart                                               # blocks per side; the UV square has nb x nb slots
    assert nb * chart == cam, "camera size must be a multiple of the chart size"
    idx = torch.arange(cam, device=device)
    by, bx = torch.meshgrid(idx // chart, idx // chart, indexing='ij')
    ly, lx = torch.meshgrid(idx % chart, idx % chart, indexing='ij')
    block = (by * nb + bx).reshape(-1)                              # [cam * cam] block id of every pixel
    yy, xx = torch.meshgrid(idx, idx, indexing='ij')
    r2 = (yy - (cam - 1) / 2.0) ** 2 + (xx - (cam - 1) / 2.0) ** 2
    fg = r2 <= fg_frac / math.pi * cam * cam
    out = torch.zeros((n_frames, cam, cam, 2), device=device, dtype=torch.float32)
    for f in range(n_frames):
        slot = torch.randperm(nb * nb, device=device, generator=g)[block].view(cam, cam)
        mode = torch.randint(0, 8, (nb * nb,), device=device, generator=g)[block].view(cam, cam)
        a, b = lx.float(), ly.float()
        swap = (mode & 1).bool()
        a, b = torch.where(swap, b, a), torch.where(swap, a, b)
        a = torch.where((mode & 2).bool(), chart - 1 - a, a)
        b = torch.where((mode & 4).bool(), chart - 1 - b, b)
        u = ((slot % nb).float() * chart + a + 0.5) / cam
        v = ((slot // nb).float() * chart + b + 0.5) / cam
        out[f] = torch.stack([u, v], -1) * fg[..., None]
    return out.half()


def synthetic_store(n_frames, uv, cam, device='cuda', seed=0, k=1, fg_frac=0.7, warp='charts'):
    """n_frames 'trainvali' samples on a cams x lights lattice; every sample's k nearest neighbours are the next k
    frames (cyclic), written the way `nn.json` holds them ({'cam', 'light'} dicts; a list when k > 1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    R = lambda *s: torch.randint(0, 256, s, device=device, generator=g, dtype=torch.uint8)
    ids = ['trainvali_%09d_C%03d_L%03d' % (i, i, i) for i in range(n_frames)]
    if warp == 'charts':
        warp = chart_warp(n_frames, cam, g, device, fg_frac)
    else:
        assert warp == 'random', warp
        warp = torch.rand((n_frames, cam, cam, 2), device=device, generator=g).half()
        warp[torch.rand((n_frames, cam, cam), device=device, generator=g) >= fg_frac] = 0
    nn = {}
    for i, id_ in enumerate(ids):
        nbrs = [{'cam': 'C%03d' % ((i + 1 + j) % n_frames), 'light': 'L%03d' % ((i + 1 + j) % n_frames)} for j in range(k)]
        nn[id_] = nbrs if k > 1 else nbrs[0]
    return {'ids': ids, 'nn': nn, 'complete': [True] * n_frames,
            'diffuse': R(n_frames, uv, uv, 3), 'rgb': R(n_frames, uv, uv, 3), 'cvis': R(n_frames, uv, uv), 'lvis': R(n_frames, uv, uv),
            'rgb_camspc': R(n_frames, cam, cam, 3), 'uv2cam': warp}
