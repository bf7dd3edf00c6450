"""CPU: the algebra behind csrc/train_fused.hip (backward of layers 0-1 without full-resolution feature tensors).
L0 is linear, so every weight gradient of L0 / level 1's stride-2 convs / the head's skip rows is a small product with
texel sums GQ, GO, H, SQ, SO, P of (raw channel x gradient).  This test forms the sums with einsum, applies the epilogue
formulas exactly as front_bwd_epilogue_kernel does, and compares with torch autograd through the UNfolded layers."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T


@pytest.mark.parametrize('n,h,w,k', [(1, 4, 8, 1), (2, 6, 8, 3)])
def test_texel_sum_identities_reproduce_autograd(n, h, w, k):
    rng = np.random.default_rng(h + k)
    R = lambda *s: torch.from_numpy(rng.standard_normal(s))                      # float64: identities, not rounding
    xq, y = R(n, h, w, 5), R(n, k, h, w, 3)                                      # [base cvis lvis], nn_rgb - nn_base per observation
    dy1q, dy1o, dpred = R(n, h // 2, w // 2, 16), R(n, k, h // 2, w // 2, 16), R(n, h, w, 3)
    W = dict(wq0=R(1, 1, 5, 16), bq0=R(16), wo0=R(1, 1, 3, 16), bo0=R(16), wqa=R(2, 2, 32, 16), woa=R(2, 2, 16, 16), wh=R(1, 1, 36, 3))
    wt = {k_: v.clone().requires_grad_(True) for k_, v in W.items()}
    q0 = xq @ wt['wq0'][0, 0] + wt['bq0']
    o0 = y @ wt['wo0'][0, 0] + wt['bo0']
    fm0 = torch.cat((q0, o0.mean(1)), -1)
    z = torch.zeros(16, dtype=torch.float64)
    s = (T.conv2d_same(fm0, wt['wqa'], z, 2) * dy1q).sum() + (fm0 @ wt['wh'][0, 0, 4:, :] * dpred).sum()
    for i in range(k):
        s = s + (T.conv2d_same(o0[:, i], wt['woa'], z, 2) * dy1o[:, i]).sum()
    ref = dict(zip(wt, torch.autograd.grad(s, list(wt.values()))))

    # texel sums (what front_bwd_kernel accumulates on the matrix cores)
    r = torch.cat((xq, y.mean(1)), -1)                                           # raw 8-vector per full-resolution texel
    taps = lambda t: torch.stack([t[:, a::2, b::2] for a in (0, 1) for b in (0, 1)], 1)    # [n, tap, h/2, w/2, c]
    GQ = torch.einsum('ntijc,nijo->tco', taps(r), dy1q)                          # [tap, 8, 16]
    GO = sum(torch.einsum('ntijc,nijo->tco', taps(y[:, i]), dy1o[:, i]) for i in range(k))   # [tap, 3, 16]
    SQ, SO = dy1q.sum((0, 1, 2)), dy1o.sum((0, 1, 2, 3))
    H, P = torch.einsum('nijc,nijo->co', r, dpred), dpred.sum((0, 1, 2))         # [8, 3], [3]

    wq0, bq0, wo0, bo0 = W['wq0'][0, 0], W['bq0'], W['wo0'][0, 0], W['bo0']
    wqa, woa, wh = W['wqa'].reshape(4, 32, 16), W['woa'].reshape(4, 16, 16), W['wh'][0, 0]
    # epilogue (front_bwd_epilogue_kernel, same index conventions)
    dwqa = torch.zeros(4, 32, 16, dtype=torch.float64)
    dwqa[:, :16] = torch.einsum('cm,tco->tmo', wq0, GQ[:, :5]) + bq0[None, :, None] * SQ[None, None, :]
    dwqa[:, 16:] = torch.einsum('cm,tco->tmo', wo0, GQ[:, 5:]) + bo0[None, :, None] * SQ[None, None, :]
    dwoa = torch.einsum('cm,tco->tmo', wo0, GO) + bo0[None, :, None] * SO[None, None, :]
    dwq0 = torch.einsum('tco,tmo->cm', GQ[:, :5], wqa[:, :16]) + H[:5] @ wh[4:20].t()
    dbq0 = torch.einsum('tmo,o->m', wqa[:, :16], SQ) + wh[4:20] @ P
    dwo0 = (torch.einsum('tco,tmo->cm', GQ[:, 5:], wqa[:, 16:]) + torch.einsum('tco,tmo->cm', GO, woa) + H[5:] @ wh[20:36].t())
    dbo0 = torch.einsum('tmo,o->m', wqa[:, 16:], SQ) + torch.einsum('tmo,o->m', woa, SO) + wh[20:36] @ P
    dwh = torch.zeros(36, 3, dtype=torch.float64)
    dwh[4:20] = wq0.t() @ H[:5] + bq0[:, None] * P[None, :]
    dwh[20:36] = wo0.t() @ H[5:] + bo0[:, None] * P[None, :]

    close = lambda a, b: np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=1e-9, atol=1e-9)
    close(dwqa.reshape(2, 2, 32, 16), ref['wqa'])
    close(dwoa.reshape(2, 2, 16, 16), ref['woa'])
    close(dwq0, ref['wq0'][0, 0]); close(dbq0, ref['bq0'])
    close(dwo0, ref['wo0'][0, 0]); close(dbo0, ref['bo0'])
    close(dwh, ref['wh'][0, 0])
