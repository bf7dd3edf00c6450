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


@pytest.mark.parametrize('n,h2,w2', [(1, 3, 4), (2, 2, 5)])
def test_last_block_backward_formulas_reproduce_autograd(n, h2, w2):
    """The per-texel formulas of csrc/train_back.hip (phases 1-3), written out with explicit index shifts:
    dv = lrelu'(v) . Wh dpred;  du[y,x] = lrelu'(u) . sum_{a,b} W1[a,b]^T dv[y+a, x+b];  dW1[a,b,o,c] = sum u[y-a,x-b,c] dv[y,x,o];
    dW2[a,b,o,c] = sum in[i,j,c] du[2i+a,2j+b,o];  d_in[i,j,c] = sum_{a,b,o} W2[a,b,o,c] du[2i+a,2j+b,o]."""
    rng = np.random.default_rng(h2 * 3 + w2)
    R = lambda *s: torch.from_numpy(rng.standard_normal(s))
    alpha = 0.3
    xin, dpred = R(n, h2, w2, 40), R(n, 2 * h2, 2 * w2, 3)
    W2, b2, W1, b1, Wh = R(2, 2, 4, 40), R(4), R(2, 2, 4, 4), R(4), R(36, 3)
    dpred[:, 0, 0, :] = 0                                                         # set_left_top_corner: that texel carries no gradient
    leaves = [t.clone().requires_grad_(True) for t in (xin, W2, b2, W1, b1, Wh)]
    u = T.leaky_relu(T.conv2d_transpose_same(leaves[0], leaves[1], leaves[2], 2), alpha)
    v = T.leaky_relu(T.conv2d_transpose_same(u, leaves[3], leaves[4], 1), alpha)
    ref = torch.autograd.grad(((v @ leaves[5][:4]) * dpred).sum(), leaves)
    u, v = u.detach(), v.detach()
    slope = lambda a: torch.where(a > 0, torch.ones_like(a), torch.full_like(a, alpha))
    h, w = 2 * h2, 2 * w2

    dv = slope(v) * (dpred @ Wh[:4].t())                                          # phase 1
    pad_br = lambda t: torch.nn.functional.pad(t, (0, 0, 0, 1, 0, 1))             # zero beyond bottom / right
    pad_tl = lambda t: torch.nn.functional.pad(t, (0, 0, 1, 0, 1, 0))             # zero above / left
    dvp, up = pad_br(dv), pad_tl(u)
    du = torch.zeros_like(u)
    dW1 = torch.zeros(2, 2, 4, 4, dtype=torch.float64)
    for a in (0, 1):
        for b in (0, 1):
            du += dvp[:, a:a + h, b:b + w] @ W1[a, b]                             # [.., o] @ [o, c]: sum_o W1[a,b,o,c] dv[y+a,x+b,o]
            ush = up[:, 1 - a:1 - a + h, 1 - b:1 - b + w]                         # u[y - a, x - b]
            dW1[a, b] = torch.einsum('nyxo,nyxc->oc', dv, ush)
    du = slope(u) * du                                                            # phase 2
    dW2 = torch.zeros(2, 2, 4, 40, dtype=torch.float64)
    d_in = torch.zeros_like(xin)
    for a in (0, 1):
        for b in (0, 1):
            sub = du[:, a::2, b::2]                                               # du[2i + a, 2j + b]
            dW2[a, b] = torch.einsum('nijo,nijc->oc', sub, xin)                   # phase 3a
            d_in += sub @ W2[a, b]                                                # phase 3b
    dWh = torch.zeros(36, 3, dtype=torch.float64)
    dWh[:4] = torch.einsum('nyxc,nyxo->co', v, dpred)

    close = lambda a, b: np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-9, atol=1e-9)
    close(d_in, ref[0]); close(dW2, ref[1]); close(du.sum((0, 1, 2)), ref[2])
    close(dW1, ref[3]); close(dv.sum((0, 1, 2)), ref[4]); close(dWh, ref[5])
