"""Seeded synthetic capture in the resident uint8 store layout (`datasets/nlt.py:Dataset`): stand-in for a decoded
dragon capture when none is on disk (bench.py, smoke tests).  Value ranges mimic what `data_gen` writes: uint8 texel
buffers, fp16 `uv2cam` maps with 30 % background pixels at (0,0) (data_gen/render.py:155, data_gen/util.py:67-70)."""
import torch


def synthetic_store(n_frames, uv, cam, device='cuda', seed=0, k=1, fg_frac=0.7):
    """n_frames 'trainvali' samples on a cams x lights lattice; every sample's k nearest neighbours are the next k
    frames (cyclic), written the way `nn.json` holds them ({'cam', 'light'} dicts; a list when k > 1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    R = lambda *s: torch.randint(0, 256, s, device=device, generator=g, dtype=torch.uint8)
    ids = ['trainvali_%09d_C%03d_L%03d' % (i, i, i) for i in range(n_frames)]
    warp = torch.rand((n_frames, cam, cam, 2), device=device, generator=g).half()
    warp[torch.rand((n_frames, cam, cam), device=device, generator=g) >= fg_frac] = 0
    nn = {}
    for i, id_ in enumerate(ids):
        nbrs = [{'cam': 'C%03d' % ((i + 1 + j) % n_frames), 'light': 'L%03d' % ((i + 1 + j) % n_frames)} for j in range(k)]
        nn[id_] = nbrs if k > 1 else nbrs[0]
    return {'ids': ids, 'nn': nn, 'complete': [True] * n_frames,
            'diffuse': R(n_frames, uv, uv, 3), 'rgb': R(n_frames, uv, uv, 3), 'cvis': R(n_frames, uv, uv), 'lvis': R(n_frames, uv, uv),
            'rgb_camspc': R(n_frames, cam, cam, 3), 'uv2cam': warp}
