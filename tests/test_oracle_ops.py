"""Cross-checks the oracle's torch-CPU primitives against independent naive NumPy loops
written from the TF/Keras/TFA documented semantics (parity with real TF is UNPINNED:
TensorFlow is not installable here -- SURVEY.md 8c).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T
from oracle import nlt_oracle as O


@pytest.mark.parametrize('k,s,h,w', [(1, 1, 5, 6), (2, 2, 6, 8), (2, 1, 5, 7), (2, 1, 1, 1), (2, 2, 2, 2)])
def test_conv_same_vs_naive(k, s, h, w):
    rng = np.random.default_rng(k * 10 + s)
    x = rng.standard_normal((2, h, w, 3)).astype(np.float32)
    wk = rng.standard_normal((k, k, 3, 4)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    got = T.conv2d_same(torch.tensor(x), torch.tensor(wk), torch.tensor(b), s).numpy()
    ref = T.conv2d_same_naive(x, wk, b, s)
    assert got.shape == ref.shape == (2, -(-h // s), -(-w // s), 4)
    np.testing.assert_allclose(got, ref, atol=1e-5)


def test_conv_k2s1_pads_bottom_right():
    # TF SAME, k=2, s=1: pad_before=0, pad_after=1 -> last row/col see zeros below/right
    x = torch.ones(1, 3, 3, 1)
    w = torch.tensor([1., 10., 100., 1000.]).reshape(2, 2, 1, 1)
    y = T.conv2d_same(x, w, torch.zeros(1), 1)[0, :, :, 0].numpy()
    np.testing.assert_array_equal(y, [[1111, 1111, 101], [1111, 1111, 101], [11, 11, 1]])


@pytest.mark.parametrize('s,h,w', [(2, 3, 4), (1, 4, 5), (1, 1, 1)])
def test_deconv_same_vs_naive(s, h, w):
    rng = np.random.default_rng(s)
    x = rng.standard_normal((2, h, w, 5)).astype(np.float32)
    wk = rng.standard_normal((2, 2, 3, 5)).astype(np.float32)      # (kh,kw,Cout,Cin)
    b = rng.standard_normal(3).astype(np.float32)
    got = T.conv2d_transpose_same(torch.tensor(x), torch.tensor(wk), torch.tensor(b), s).numpy()
    ref = T.conv2d_transpose_same_naive(x, wk, b, s)
    assert got.shape == ref.shape == (2, h * s, w * s, 3)
    np.testing.assert_allclose(got, ref, atol=1e-5)


def test_deconv_is_gradient_of_conv():
    # Conv2DTranspose(padding='same') == d/dx of the SAME conv with the same kernel array
    rng = np.random.default_rng(3)
    for s in (1, 2):
        x = torch.tensor(rng.standard_normal((1, 4, 6, 3)).astype(np.float32), requires_grad=True)
        wk = torch.tensor(rng.standard_normal((2, 2, 3, 5)).astype(np.float32))   # conv HWIO: in=3,out=5
        y = T.conv2d_same(x, wk, torch.zeros(5), s)
        dy = torch.tensor(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        (gx,) = torch.autograd.grad(y, x, dy)
        # same array read as (kh,kw,Cout=3,Cin=5) is the transposed-conv kernel
        tx = T.conv2d_transpose_same(dy, wk, torch.zeros(3), s)
        np.testing.assert_allclose(gx.numpy(), tx.numpy(), atol=1e-5)


def test_resampler_vs_naive_and_integer_coords():
    rng = np.random.default_rng(0)
    data = rng.random((2, 6, 7, 3), dtype=np.float32)
    warp = (rng.random((2, 5, 4, 2), dtype=np.float32) * np.float32([9, 8]) - 1.5).astype(np.float32)
    warp[0, 0, 0] = (0, 0); warp[0, 0, 1] = (6.0, 5.0); warp[0, 0, 2] = (7.0, 2.0); warp[0, 0, 3] = (-1.0, 2.0)
    warp[1, 0, 0] = (-0.5, -0.5); warp[1, 0, 1] = (6.5, 5.5); warp[1, 0, 2] = (3.0, 2.0)
    ref = T.resampler_naive(data, warp)
    got = T.resampler(torch.tensor(data), torch.tensor(warp)).numpy()
    np.testing.assert_allclose(got, ref, atol=1e-6)
    np.testing.assert_array_equal(ref[1, 0, 2], data[1, 2, 3])     # integer coords == index gather
    np.testing.assert_array_equal(ref[0, 0, 0], data[0, 0, 0])
    np.testing.assert_array_equal(ref[0, 0, 2], 0)                 # x == W -> outside
    np.testing.assert_array_equal(ref[0, 0, 3], 0)                 # x == -1 -> outside
    np.testing.assert_allclose(ref[1, 0, 0], 0.25 * data[1, 0, 0], atol=1e-7)  # implicit zero border
    fx, fy, inside = T.resampler_indices(warp, 6, 7)
    assert fx.dtype == np.int32 and fx[1, 0, 2] == 3 and fy[1, 0, 2] == 2 and not inside[0, 0, 2]


def test_resampler_identity_warp_is_noop():
    rng = np.random.default_rng(1)
    data = rng.random((1, 8, 8, 3), dtype=np.float32)
    jj, ii = np.meshgrid(np.arange(8, dtype=np.float32), np.arange(8, dtype=np.float32))
    warp = np.stack((jj, ii), -1)[None]
    np.testing.assert_array_equal(T.resampler_naive(data, warp), data)


def test_resampler_grad_is_scatter_add():
    rng = np.random.default_rng(2)
    data = torch.tensor(rng.random((1, 4, 4, 2), dtype=np.float32), requires_grad=True)
    warp = torch.tensor([[[[1.25, 2.5], [0.0, 0.0], [3.5, 3.5]]]])
    out = T.resampler(data, warp)
    (g,) = torch.autograd.grad(out.sum(), data)
    g = g[0, :, :, 0].numpy()
    exp = np.zeros((4, 4), np.float32)
    exp[2, 1] += .75 * .5; exp[3, 2] += .25 * .5; exp[3, 1] += .75 * .5; exp[2, 2] += .25 * .5
    exp[0, 0] += 1.0
    exp[3, 3] += .25                                               # other 3 corners fall outside
    np.testing.assert_allclose(g, exp, atol=1e-6)


@pytest.mark.parametrize('oh,ow', [(8, 8), (4, 6), (16, 12), (5, 7)])
def test_resize_vs_naive(oh, ow):
    rng = np.random.default_rng(oh)
    x = rng.random((2, 8, 8, 3), dtype=np.float32)
    got = T.resize_bilinear(torch.tensor(x), oh, ow).numpy()
    ref = T.resize_bilinear_naive(x, oh, ow)
    np.testing.assert_allclose(got, ref, atol=1e-6)


def test_gen_feat_n_and_layers():       # nlt/util/net.py docstring + SURVEY a-G
    assert O.gen_feat_n(16, 256) == [16, 32, 64, 128, 256, 256, 128, 64, 32, 16, 8, 4, 3]
    assert O.gen_feat_n(16, 1024) == [16, 32, 64, 128, 256, 512, 1024, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 3]
    assert O.gen_feat_n(8, 64) == [8, 16, 32, 64, 64, 32, 16, 8, 4, 3]
    layers, is_c, _ = O.build_layers(16, 256)
    assert len(layers) == 14 and sum(is_c) == 7
    layers, is_c, _ = O.build_layers(16, 1024)
    assert len(layers) == 18 and sum(is_c) == 9
    with pytest.raises(AssertionError):
        O.build_layers(8, 64)           # depth0 must be 16: resolution does not return


def test_param_count_and_in_channels():
    m = O.OracleModel(depth=256, uvh=64, uvw=64, imh=64, imw=64)
    assert sum(p.numel() for p in m.parameters()) == 3368071
    q_in, _ = O._layer_in_channels(m.layers, m.is_contracting, 5, 3)
    assert q_in == [5, 32, 32, 64, 128, 256, 512, 1024, 640, 320, 160, 80, 40, 36]   # SURVEY 8a plan


def test_parse_loss():                  # nlt/models/base.py:63-77
    assert O.parse_loss_and_weight('1e+0lpips') == ('lpips', 1.0)
    assert O.parse_loss_and_weight('barron') == ('barron', 1.0)
    assert O.parse_loss_and_weight('10barron') == ('barron', 10.0)


def test_model_forward_and_train_step_small():
    m = O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32, loss='l2', seed=1)
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=5)
    pred_c, gt_c, kw, vis = m.call(batch, 'train', nn_list=nn)
    assert pred_c.shape == (2, 32, 32, 3) and gt_c.shape == (2, 32, 32, 3)
    assert vis['pred'].shape == (2, 64, 64, 3)
    assert torch.all(vis['pred'][:, 0, 0, :] == 0)
    with pytest.raises(ValueError):
        m.call(batch, 'bogus')
    opt = O.KerasAdamAMSGrad(m.parameters(), 1e-3)
    before = [p.detach().clone() for p in m.parameters()]
    l0, grads = O.train_step(m, opt, batch, global_bs=2, nn_list=nn)
    assert all(torch.isfinite(g).all() for g in grads)
    # first Adam step moves every touched weight by ~lr (|m/sqrt(v)|=1 at t=1)
    delta = max((a - b.detach()).abs().max().item() for a, b in zip(before, m.parameters()))
    assert 0.5e-3 < delta <= 1.001e-3
    l1, _ = O.train_step(m, opt, batch, global_bs=2, nn_list=nn)
    assert l1 < l0
