"""-m gpu: the training forms of the fused ends (csrc/fused.hip *_train entry points, csrc/train_fused.hip) against the
CPU oracle primitives: saved activations of the forward launches, and the weight gradients of L0 / level 1's stride-2
convs / the head's skip rows from nlt_front_backward against torch autograd through the UNfolded layers (oracle/tf_ops).
Tolerance: rel-L2 <= 1e-5 for activations, <= 2e-5 for the texel-sum gradients (fp32 summation order only)."""
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from oracle import tf_ops as T
from gpu_util import rel_l2

pytestmark = pytest.mark.gpu


def _rand_weights(rng):
    G = lambda *s: torch.from_numpy((rng.random(s, dtype=np.float32) - 0.5) * 0.8)
    return dict(wq0=G(1, 1, 5, 16), bq0=G(16), wo0=G(1, 1, 3, 16), bo0=G(16), wqa=G(2, 2, 32, 16), bqa=G(16), wqb=G(2, 2, 16, 16),
                bqb=G(16), woa=G(2, 2, 16, 16), boa=G(16), wob=G(2, 2, 16, 16), bob=G(16), wh=G(1, 1, 36, 3), bh=G(3))


ORDER = ('wq0', 'bq0', 'wo0', 'bo0', 'wqa', 'bqa', 'wqb', 'bqb', 'woa', 'boa', 'wob', 'bob', 'wh', 'bh')


@pytest.mark.parametrize('n,h,w,k', [(1, 16, 16, 1), (2, 24, 40, 3), (1, 64, 128, 4), (1, 8, 8, 6)])
def test_front_forward_train_keeps_the_stride2_outputs(n, h, w, k):
    rng = np.random.default_rng(h + w + k)
    U = lambda *s: torch.from_numpy(rng.random(s, dtype=np.float32))
    P = _rand_weights(rng)
    base, cvis, lvis, nn_rgb, nn_base = U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1), U(n, k, h, w, 3), U(n, k, h, w, 3)
    lr = lambda x: T.leaky_relu(x, 0.3)
    q0 = torch.cat((base, cvis, lvis), -1) @ P['wq0'][0, 0] + P['bq0']
    o0 = (nn_rgb - nn_base) @ P['wo0'][0, 0] + P['bo0']
    q_ref = lr(T.conv2d_same(torch.cat((q0, o0.mean(1)), -1), P['wqa'], P['bqa'], 2))
    o_ref = torch.stack([lr(T.conv2d_same(o0[:, i], P['woa'], P['boa'], 2)) for i in range(k)], 1)
    fm1_ref = torch.cat((lr(T.conv2d_same(q_ref, P['wqb'], P['bqb'], 1)),
                         torch.stack([lr(T.conv2d_same(o_ref[:, i], P['wob'], P['bob'], 1)) for i in range(k)], 1).mean(1)), -1)
    dev = lambda a: a.cuda().contiguous()
    blob = C.front_pack_weights(*[dev(P[k_]) for k_ in ORDER])
    nan = lambda *s: torch.full(s, float('nan'), device='cuda')
    fm1, obs1, skip3 = nan(n, h // 2, w // 2, 32), nan(n, k, h // 2, w // 2, 16), nan(n, h, w, 3)
    qt, ot = nan(n, h // 2, w // 2, 16), nan(n, k, h // 2, w // 2, 16)
    C.front_forward_train(dev(base), dev(cvis), dev(lvis), dev(nn_rgb), dev(nn_base), n, k, h, w, blob, True, 0.3,
                          fm1, obs1, skip3, qt, ot)
    torch.cuda.synchronize()
    for got, ref in ((qt, q_ref), (ot, o_ref), (fm1, fm1_ref)):
        assert not torch.isnan(got).any()
        assert rel_l2(got.cpu(), ref) <= 1e-5


@pytest.mark.parametrize('n,h2,w2', [(1, 8, 16), (2, 12, 20), (1, 3, 5), (1, 32, 64)])
def test_back_forward_train_keeps_the_last_blocks_maps(n, h2, w2):
    rng = np.random.default_rng(h2 * 3 + w2)
    U = lambda *s: torch.from_numpy(rng.random(s, dtype=np.float32) - 0.5)
    x, fm1, skip3 = U(n, h2, w2, 8), U(n, h2, w2, 32), U(n, 2 * h2, 2 * w2, 3)
    w_s2, b_s2, w_s1, b_s1, wh = U(2, 2, 4, 40), U(4), U(2, 2, 4, 4), U(4), U(1, 1, 36, 3)
    lr = lambda t: T.leaky_relu(t, 0.3)
    u_ref = lr(T.conv2d_transpose_same(torch.cat((x, fm1), -1), w_s2, b_s2, 2))
    v_ref = lr(T.conv2d_transpose_same(u_ref, w_s1, b_s1, 1))
    p_ref = T.set_left_top_corner(v_ref @ wh[0, 0, :4] + skip3, 0)
    dev = lambda a: a.cuda().contiguous()
    nan = lambda *s: torch.full(s, float('nan'), device='cuda')
    pred, u, v = nan(n, 2 * h2, 2 * w2, 3), nan(n, 2 * h2, 2 * w2, 4), nan(n, 2 * h2, 2 * w2, 4)
    C.back_forward_train(dev(x), dev(fm1), dev(skip3), n, h2, w2, dev(w_s2), dev(b_s2), dev(w_s1), dev(b_s1), dev(wh), 0.3,
                         pred, u, v)
    torch.cuda.synchronize()
    for got, ref in ((u, u_ref), (v, v_ref), (pred, p_ref)):
        assert not torch.isnan(got).any()
        assert rel_l2(got.cpu(), ref) <= 1e-5


@pytest.mark.parametrize('n,h,w,k', [(1, 8, 8, 1), (2, 24, 40, 3), (1, 64, 128, 4), (3, 32, 16, 2), (1, 256, 512, 1), (2, 32, 64, 2),
                                     (1, 16, 32, 3), (3, 2, 32, 1)])
def test_front_backward_matches_autograd_through_the_unfolded_layers(n, h, w, k):
    rng = np.random.default_rng(h * 5 + w + k)
    U = lambda *s: torch.from_numpy(rng.random(s, dtype=np.float32))
    S = lambda *s: torch.from_numpy((rng.random(s, dtype=np.float32) - 0.5))
    P = _rand_weights(rng)
    base, cvis, lvis, nn_rgb, nn_base = U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1), U(n, k, h, w, 3), U(n, k, h, w, 3)
    dy1q, dy1o, dpred = S(n, h // 2, w // 2, 16), S(n, k, h // 2, w // 2, 16), S(n, h, w, 3)
    names = ('wq0', 'bq0', 'wo0', 'bo0', 'wqa', 'woa', 'wh')
    g = dpred.clone(); g[:, 0, 0, :] = 0
    with torch.enable_grad():
        wt = {k_: P[k_].clone().requires_grad_(True) for k_ in names}
        q0 = torch.cat((base, cvis, lvis), -1) @ wt['wq0'][0, 0] + wt['bq0']
        o0 = (nn_rgb - nn_base) @ wt['wo0'][0, 0] + wt['bo0']
        fm0 = torch.cat((q0, o0.mean(1)), -1)
        zero = torch.zeros(16)
        s = (T.conv2d_same(fm0, wt['wqa'], zero, 2) * dy1q).sum() + (fm0 @ wt['wh'][0, 0, 4:, :] * g).sum()
        for i in range(k):
            s = s + (T.conv2d_same(o0[:, i], wt['woa'], zero, 2) * dy1o[:, i]).sum()
        ref = dict(zip(names, torch.autograd.grad(s, [wt[k_] for k_ in names])))
    ref['bqa'] = dy1q.reshape(-1, 16).sum(0)
    ref['boa'] = dy1o.reshape(-1, 16).sum(0)
    dev = lambda a: a.cuda().contiguous()
    gnames = ('wq0', 'bq0', 'wo0', 'bo0', 'wqa', 'bqa', 'woa', 'boa', 'wh')
    init = {k_: torch.from_numpy(rng.random(tuple(P[k_].shape), dtype=np.float32)) for k_ in gnames}   # += : start from non-zero
    grads = {k_: dev(init[k_]) for k_ in gnames}
    C.front_backward(dev(base), dev(cvis), dev(lvis), dev(nn_rgb), dev(nn_base), n, k, h, w, dev(dy1q), dev(dy1o), dev(dpred),
                     tuple(dev(P[k_]) for k_ in names), tuple(grads[k_] for k_ in gnames))
    torch.cuda.synchronize()
    for k_ in gnames:
        got = grads[k_].cpu() - init[k_]
        if k_ == 'wh':
            assert float(got[0, 0, :4].abs().max()) <= 1e-6              # the decoder rows belong to the back side
            got, r = got[0, 0, 4:], ref[k_][0, 0, 4:]
        else:
            r = ref[k_]
        assert rel_l2(got, r) <= 3e-5, (k_, rel_l2(got, r))
    # run-to-run determinism (two-pass reductions, no atomics)
    again = {k_: dev(init[k_]) for k_ in gnames}
    C.front_backward(dev(base), dev(cvis), dev(lvis), dev(nn_rgb), dev(nn_base), n, k, h, w, dev(dy1q), dev(dy1o), dev(dpred),
                     tuple(dev(P[k_]) for k_ in names), tuple(again[k_] for k_ in gnames))
    torch.cuda.synchronize()
    assert all(torch.equal(again[k_], grads[k_]) for k_ in gnames)


@pytest.mark.parametrize('n,h2,w2', [(1, 8, 16), (2, 12, 20), (1, 3, 5), (1, 32, 64), (3, 40, 24), (2, 128, 256)])
def test_back_backward_matches_autograd_through_the_last_block(n, h2, w2):
    rng = np.random.default_rng(h2 * 11 + w2)
    S = lambda *s: torch.from_numpy(rng.random(s, dtype=np.float32) - 0.5)
    x, fm1, dpred = S(n, h2, w2, 8), S(n, h2, w2, 32), S(n, 2 * h2, 2 * w2, 3)
    P = dict(w_s2=S(2, 2, 4, 40), b_s2=S(4), w_s1=S(2, 2, 4, 4), b_s1=S(4), wh=S(1, 1, 36, 3))
    g = dpred.clone(); g[:, 0, 0, :] = 0
    with torch.enable_grad():
        xin = torch.cat((x, fm1), -1).requires_grad_(True)
        wt = {k_: a.clone().requires_grad_(True) for k_, a in P.items()}
        u = T.leaky_relu(T.conv2d_transpose_same(xin, wt['w_s2'], wt['b_s2'], 2), 0.3)
        v = T.leaky_relu(T.conv2d_transpose_same(u, wt['w_s1'], wt['b_s1'], 1), 0.3)
        s = ((v @ wt['wh'][0, 0, :4]) * g).sum()
        names = ('w_s2', 'b_s2', 'w_s1', 'b_s1', 'wh')
        gr = torch.autograd.grad(s, [xin] + [wt[k_] for k_ in names])
    ref = dict(zip(names, gr[1:]))
    ref['bh'] = g.reshape(-1, 3).sum(0)
    dev = lambda a: a.detach().cuda().contiguous()
    init = {k_: torch.from_numpy(rng.random(tuple(ref[k_].shape), dtype=np.float32)) for k_ in ref}
    grads = {k_: dev(init[k_]) for k_ in ref}
    dx = torch.full((n, h2, w2, 8), float('nan'), device='cuda')
    dfm1 = torch.full((n, h2, w2, 32), float('nan'), device='cuda')
    run = lambda G: C.back_backward(dev(x), dev(fm1), dev(u), dev(v), dev(dpred), n, h2, w2, dev(P['w_s2']), dev(P['w_s1']),
                                    dev(P['wh']), 0.3, dx, dfm1, G['w_s2'], G['b_s2'], G['w_s1'], G['b_s1'], G['wh'], G['bh'])
    run(grads)
    torch.cuda.synchronize()
    assert not torch.isnan(dx).any() and not torch.isnan(dfm1).any()
    dx_ref = gr[0][..., :8] * torch.where(x > 0, 1.0, 0.3)              # handed back w.r.t. the producer's pre-activation
    assert rel_l2(dx.cpu(), dx_ref) <= 1e-5 and rel_l2(dfm1.cpu(), gr[0][..., 8:]) <= 1e-5
    for k_ in ref:
        got = grads[k_].cpu() - init[k_]
        r = ref[k_]
        if k_ == 'wh':
            assert float(got[0, 0, 4:].abs().max()) <= 1e-6              # the skip rows belong to the front side
            got, r = got[0, 0, :4], r[0, 0, :4]
        assert rel_l2(got, r) <= 3e-5, (k_, rel_l2(got, r))
    again = {k_: dev(init[k_]) for k_ in ref}
    run(again)
    torch.cuda.synchronize()
    assert all(torch.equal(again[k_], grads[k_]) for k_ in ref)
