"""-m gpu: backward / loss / optimizer kernels against torch-CPU autograd over the oracle's primitives."""
import os
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from oracle import tf_ops as T
from oracle import barron as B
from oracle import nlt_oracle as O

pytestmark = pytest.mark.gpu
d = lambda a: None if a is None else torch.tensor(a).cuda()
MODES = {C.CONV1X1: (1, 1, False), C.CONV_K2S2: (2, 2, False), C.CONV_K2S1: (2, 1, False),
         C.DECONV_K2S2: (2, 2, True), C.DECONV_K2S1: (2, 1, True)}


def wgrad_case(mode, n, h, w, c0, c1, cout, algo, pad0=0, pad1=0, padp=0, seed=0):
    rng = np.random.default_rng(seed)
    k, s, tr = MODES[mode]
    cin = c0 + c1
    x0 = rng.standard_normal((n, h, w, c0 + pad0)).astype(np.float32)
    x1 = rng.standard_normal((n, h, w, c1 + pad1)).astype(np.float32) if c1 else None
    x = x0[..., :c0] if not c1 else np.concatenate((x0[..., :c0], x1[..., :c1]), -1)
    wshape = (k, k, cout, cin) if tr else (k, k, cin, cout)
    wz = torch.zeros(wshape, requires_grad=True); bz = torch.zeros(cout, requires_grad=True)
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    y = f(torch.tensor(x), wz, bz, s)
    oh, ow = y.shape[1:3]
    dp = rng.standard_normal((n, oh, ow, cout + padp)).astype(np.float32)
    gw, gb = torch.autograd.grad(y, (wz, bz), torch.tensor(dp[..., :cout]))
    dw = torch.zeros(wshape, device='cuda'); db = torch.zeros(cout, device='cuda')
    if algo == 'narrow':
        C.conv_backward_weights_narrow(mode, d(x0), c0, c0 + pad0, d(x1), c1, c1 + pad1, n, h, w, d(dp), cout + padp, cout, dw, db)
    elif algo == 'tiled':
        C.conv_backward_weights_tiled(mode, d(x0), c0, c0 + pad0, d(x1), c1, c1 + pad1, n, h, w, d(dp), cout + padp, cout, dw, db)
    else:
        C.conv_backward_weights(mode, d(x0), c0, c0 + pad0, d(x1), c1, c1 + pad1, n, h, w, d(dp), cout + padp, cout, dw, db,
                                algo=algo)
    torch.cuda.synchronize()
    sw, sb = max(gw.abs().max().item(), 1.0), max(gb.abs().max().item(), 1.0)
    np.testing.assert_allclose(dw.cpu().numpy(), gw.numpy(), atol=3e-5 * sw)
    np.testing.assert_allclose(db.cpu().numpy(), gb.numpy(), atol=3e-5 * sb)


@pytest.mark.parametrize('mode', list(MODES))
@pytest.mark.parametrize('algo', [C.ALGO_DIRECT, C.ALGO_MFMA, 'tiled'])
def test_wgrad_modes(mode, algo):
    wgrad_case(mode, 2, 6, 10, 16, 0, 16, algo, seed=mode)


@pytest.mark.parametrize('mode', list(MODES))
@pytest.mark.parametrize('algo', [C.ALGO_MFMA, 'tiled'])
def test_wgrad_mfma_dual_source_slices_ragged(mode, algo):
    wgrad_case(mode, 3, 6, 10, 24, 40, 48, algo, pad0=8, pad1=4, padp=16, seed=10 + mode)


@pytest.mark.parametrize('mode,c0,c1,cout', [(C.DECONV_K2S2, 8, 32, 4), (C.DECONV_K2S1, 4, 0, 4), (C.CONV_K2S2, 512, 0, 256),
                                             (C.DECONV_K2S2, 512, 512, 128), (C.CONV1X1, 4, 32, 12)])
@pytest.mark.parametrize('algo', [C.ALGO_MFMA, 'tiled'])
def test_wgrad_mfma_shapes(mode, c0, c1, cout, algo):
    hw = 8 if algo == 'tiled' else 4          # the tiled kernel needs >= 4 texels per GEMM grid row (4 after the stride-2 conv)
    wgrad_case(mode, 2, hw, hw, c0, c1, cout, algo, seed=20)


NARROW = [(C.CONV_K2S1, 16, 0, 16), (C.CONV_K2S1, 32, 0, 32), (C.CONV_K2S2, 32, 0, 32), (C.CONV_K2S2, 16, 0, 32),
          (C.CONV_K2S2, 32, 0, 16), (C.DECONV_K2S1, 8, 0, 8), (C.DECONV_K2S1, 16, 0, 16), (C.DECONV_K2S1, 4, 0, 4),
          (C.DECONV_K2S2, 16, 64, 8), (C.DECONV_K2S2, 8, 32, 4), (C.CONV_K2S1, 5, 0, 7), (C.CONV_K2S2, 3, 6, 12),
          (C.DECONV_K2S1, 32, 0, 32)]


@pytest.mark.parametrize('mode,c0,c1,cout', NARROW)
def test_wgrad_narrow_layers(mode, c0, c1, cout):
    """csrc/wgrad_narrow.hip: the released net's 8/16/32-column layers (+ odd channel counts), ragged leading dims."""
    wgrad_case(mode, 2, 12, 20, c0, c1, cout, 'narrow', pad0=4, pad1=8 if c1 else 0, padp=4, seed=60 + cout)
    wgrad_case(mode, 1, 8, 8, c0, c1, cout, 'narrow', seed=61)               # a single partial step per wave


@pytest.mark.parametrize('mode', [C.CONV_K2S1, C.DECONV_K2S1])
@pytest.mark.parametrize('c', [16, 32])
def test_wgrad_lds_tiled_stride1_layers(mode, c, monkeypatch):
    """r05, wgrad_s1t_kernel (the 16 / 32-channel stride-1 layers through LDS tiles): many tiles per persistent workgroup, widths
    that are not a multiple of the tile, heights that are not a multiple of 4, dP as a channel slice of a wider map (the
    interleaved [query | observation-mean] gradient), and agreement with the row-walking kernel it replaces."""
    wgrad_case(mode, 3, 38, 200, c, 0, c, 'narrow', padp=c, seed=70 + c)            # dP leading dimension 2c, 3 x 10 x 4|7 tiles
    wgrad_case(mode, 2, 5, 9, c, 0, c, 'narrow', pad0=4, seed=71)                    # one clipped tile, ragged X rows
    x = torch.randn(5, 260, 132, c, device='cuda'); dp = torch.randn(5, 260, 132, 2 * c, device='cuda')   # > 768 tiles
    res = []
    for form in ('1', '0'):
        dw = torch.zeros((2, 2, c, c), device='cuda'); db = torch.zeros(c, device='cuda')
        monkeypatch.setenv('NLT_WGRAD_S1T', form)                                # '0': the row-walking kernel
        C.conv_backward_weights_narrow(mode, x, c, c, None, 0, 0, 5, 260, 132, dp, 2 * c, c, dw, db)
        res.append((dw.cpu().numpy(), db.cpu().numpy()))
    scale = np.abs(res[1][0]).max()
    np.testing.assert_allclose(res[0][0], res[1][0], atol=2e-5 * scale)
    np.testing.assert_allclose(res[0][1], res[1][1], atol=2e-5 * np.abs(res[1][1]).max())


def test_wgrad_narrow_many_row_slices_is_deterministic_accumulates_and_rejects_wide_layers():
    wgrad_case(C.CONV_K2S1, 2, 128, 96, 16, 0, 16, 'narrow', seed=62)        # 24576 rows -> ~100 row slices
    wgrad_case(C.DECONV_K2S2, 2, 64, 64, 16, 64, 8, 'narrow', seed=63)
    x = torch.randn(2, 80, 40, 32, device='cuda'); dp = torch.randn(2, 80, 40, 32, device='cuda')
    outs = []
    for _ in range(2):
        dw = torch.ones(2, 2, 32, 32, device='cuda'); db = torch.ones(32, device='cuda')
        C.conv_backward_weights_narrow(C.CONV_K2S1, x, 32, 32, None, 0, 0, 2, 80, 40, dp, 32, 32, dw, db)
        outs.append((dw.cpu().numpy(), db.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])     # bit-identical run to run
    dw0 = torch.zeros(2, 2, 32, 32, device='cuda'); db0 = torch.zeros(32, device='cuda')
    C.conv_backward_weights_narrow(C.CONV_K2S1, x, 32, 32, None, 0, 0, 2, 80, 40, dp, 32, 32, dw0, db0)
    np.testing.assert_allclose(outs[0][0] - 1, dw0.cpu().numpy(), atol=3e-4)
    assert not C.wgrad_narrow_supported(C.CONV_K2S1, 64, 0, 2, 16, 16, 64)   # 64 columns: wgrad_tile's job
    assert not C.wgrad_narrow_supported(C.CONV_K2S1, 64, 0, 2, 16, 16, 16)   # K = 256 > 128
    assert C.wgrad_narrow_supported(C.CONV_K2S2, 32, 0, 2, 16, 16, 32)


def test_wgrad_tiled_many_row_slices_is_deterministic_and_accumulates():
    # 2*64*96 = 12288 rows -> hundreds of row slices through the workspace
    wgrad_case(C.CONV_K2S1, 2, 64, 96, 16, 0, 16, 'tiled', seed=40)
    wgrad_case(C.DECONV_K2S2, 2, 48, 32, 8, 32, 4, 'tiled', padp=4, seed=41)
    x = torch.randn(2, 40, 40, 32, device='cuda'); dp = torch.randn(2, 40, 40, 32, device='cuda')
    outs = []
    for _ in range(2):
        dw = torch.ones(2, 2, 32, 32, device='cuda'); db = torch.ones(32, device='cuda')
        C.conv_backward_weights_tiled(C.CONV_K2S1, x, 32, 32, None, 0, 0, 2, 40, 40, dp, 32, 32, dw, db)
        outs.append((dw.cpu().numpy(), db.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])     # bit-identical run to run
    dw0 = torch.zeros(2, 2, 32, 32, device='cuda'); db0 = torch.zeros(32, device='cuda')
    C.conv_backward_weights_tiled(C.CONV_K2S1, x, 32, 32, None, 0, 0, 2, 40, 40, dp, 32, 32, dw0, db0)
    np.testing.assert_allclose(outs[0][0] - 1, dw0.cpu().numpy(), atol=2e-4)
    with pytest.raises(C.NLTError):                      # 5 input channels: not a multiple of 4 -> first-generation direct path only
        C.conv_backward_weights_tiled(C.CONV1X1, torch.zeros(1, 4, 4, 5, device='cuda'), 5, 5, None, 0, 0, 1, 4, 4,
                                      torch.zeros(1, 4, 4, 16, device='cuda'), 16, 16, torch.zeros(1, 1, 5, 16, device='cuda'),
                                      torch.zeros(16, device='cuda'))


def test_wgrad_direct_odd_channels_and_accumulation():
    wgrad_case(C.CONV1X1, 2, 5, 7, 5, 0, 16, C.ALGO_DIRECT, seed=30)
    wgrad_case(C.CONV_K2S1, 1, 4, 4, 3, 2, 7, C.ALGO_DIRECT, seed=31)
    # accumulation into a non-zero buffer
    x = torch.randn(1, 4, 4, 16, device='cuda'); dp = torch.randn(1, 4, 4, 16, device='cuda')
    dw = torch.ones(2, 2, 16, 16, device='cuda'); db = torch.ones(16, device='cuda')
    C.conv_backward_weights(C.CONV_K2S1, x, 16, 16, None, 0, 0, 1, 4, 4, dp, 16, 16, dw, db)
    dw2 = torch.zeros_like(dw); db2 = torch.zeros_like(db)
    C.conv_backward_weights(C.CONV_K2S1, x, 16, 16, None, 0, 0, 1, 4, 4, dp, 16, 16, dw2, db2)
    np.testing.assert_allclose((dw - 1).cpu().numpy(), dw2.cpu().numpy(), atol=1e-5)
    np.testing.assert_allclose((db - 1).cpu().numpy(), db2.cpu().numpy(), atol=1e-5)


@pytest.mark.parametrize('mode', [C.CONV_K2S2, C.CONV_K2S1, C.DECONV_K2S2, C.DECONV_K2S1])
def test_dgrad_is_adjoint_mode_on_same_kernel(mode):
    """backward-data = nlt_conv_forward in the adjoint family on the SAME Keras array."""
    from nlt_amd.networks.elements import Conv2D
    rng = np.random.default_rng(mode)
    k, s, tr = MODES[mode]
    cin, cout, n, h, w = 24, 16, 2, 4, 6
    wk = rng.standard_normal((2, 2, cout, cin) if tr else (2, 2, cin, cout)).astype(np.float32)
    x = torch.tensor(rng.standard_normal((n, h, w, cin)).astype(np.float32), requires_grad=True)
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    y = f(x, torch.tensor(wk), torch.zeros(cout), s)
    dp = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    (gx,) = torch.autograd.grad(y, x, torch.tensor(dp))
    layer = Conv2D(cout, 2, s, transpose=tr)
    layer.set_weights(wk, np.zeros(cout, np.float32))
    oh, ow = y.shape[1:3]
    zb = torch.zeros(64, device='cuda')
    for lo, hi in ((0, cin), (0, 8), (8, 24)):
        packed, ks = layer.packed_adjoint(lo, hi)
        out = torch.empty(n, h, w, hi - lo, device='cuda')
        C.conv_forward(layer.ADJOINT[mode], d(dp), cout, cout, None, 0, 0, n, oh, ow, ks, packed, zb, hi - lo, out,
                       hi - lo, act=False, algo=C.ALGO_MFMA)
        np.testing.assert_allclose(out.cpu().numpy(), gx.numpy()[..., lo:hi], atol=3e-5 * gx.abs().max().item())


@pytest.mark.parametrize('mode,cin,cout,h,w,tn', [(C.CONV_K2S1, 64, 32, 19, 37, 32), (C.CONV_K2S1, 128, 64, 16, 32, 64),
                                                  (C.DECONV_K2S1, 64, 48, 9, 20, 32), (C.DECONV_K2S2, 96, 32, 10, 18, 32),
                                                  (C.DECONV_K2S2, 128, 16, 8, 16, 64), (C.CONV_K2S1, 32, 256, 8, 8, 32),
                                                  (C.CONV_K2S2, 64, 32, 20, 36, 32), (C.CONV_K2S2, 128, 64, 16, 32, 64),
                                                  (C.CONV_K2S2, 32, 96, 12, 40, 64)])
def test_backward_data_on_the_lds_tiled_kernel(mode, cin, cout, h, w, tn):
    """nlt_conv_tile_backward_data: the gradient w.r.t. a conv's input channels [lo, hi) on csrc/conv_tile.hip -- the transposed
    k2s1 mode (halo on the top / left) for Conv2D k2s1 layers, the k2s1 / k2s2 conv modes for Conv2DTranspose layers -- read in
    place from the layer's Keras array, against torch autograd; plain, with the producer's LeakyReLU' and accumulated."""
    from nlt_amd.networks.elements import Conv2D
    rng = np.random.default_rng(mode * 100 + cin)
    k, s, tr = MODES[mode]
    n = 2
    R = lambda *shape: rng.standard_normal(shape).astype(np.float32)
    wk = R(*((2, 2, cout, cin) if tr else (2, 2, cin, cout)))
    x = torch.tensor(R(n, h, w, cin), requires_grad=True)
    y = (T.conv2d_transpose_same if tr else T.conv2d_same)(x, torch.tensor(wk), torch.zeros(cout), s)
    dp = R(*y.shape)
    (gx,) = torch.autograd.grad(y, x, torch.tensor(dp))
    layer = Conv2D(cout, 2, s, transpose=tr)
    layer.set_weights(wk, np.zeros(cout, np.float32))
    adj = layer.ADJOINT[mode]
    oh, ow = y.shape[1:3]
    tol = 3e-5 * float(gx.abs().max())
    if adj == C.DECONV_K2S2:                                    # N = 4 (hi - lo) columns: slices in multiples of 16 channels
        slices = [(0, cin), (cin - 16, cin)] + ([(16, 48)] if cin >= 48 else [])
    else:
        slices = [(0, cin), (cin - tn, cin), (tn, 2 * tn) if cin >= 2 * tn else (0, tn)]
    for lo, hi in slices:
        ld = hi - lo + 4
        existing, ymask = R(n, h, w, ld), R(n, h, w, hi - lo)
        packed = layer.packed_adjoint_tile(lo, hi, tn)
        out = d(existing)
        C.conv_tile_backward_data(adj, d(dp), cout, cout, n, oh, ow, packed, hi - lo, tn, out, ld)
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy()[..., :hi - lo], gx.numpy()[..., lo:hi], atol=tol)
        np.testing.assert_array_equal(out.cpu().numpy()[..., hi - lo:], existing[..., hi - lo:])      # the slice only
        out = d(existing)
        C.conv_tile_backward_data(adj, d(dp), cout, cout, n, oh, ow, packed, hi - lo, tn, out, ld, mask_src=d(ymask), ldm=hi - lo,
                                  mask_alpha=0.3, accumulate=True)
        torch.cuda.synchronize()
        ref = (gx[..., lo:hi] + torch.tensor(existing[..., :hi - lo])) * torch.where(torch.tensor(ymask) > 0, 1.0, 0.3)
        np.testing.assert_allclose(out.cpu().numpy()[..., :hi - lo], ref.numpy(), atol=2 * tol)


@pytest.mark.parametrize('c,cout,h,w,tn,partial', [(16, 32, 16, 32, 64, True), (32, 64, 18, 44, 32, True), (64, 96, 8, 16, 64, False)])
def test_level_split_epilogue_on_the_lds_tiled_kernel(c, cout, h, w, tn, partial):
    """The transposed k2s2 mode of csrc/conv_tile.hip with the level-split epilogue (query half x LeakyReLU' into dfm[l],
    observation half finished into dobs) against nlt_conv_backward_data on the register-tiled kernel, same inputs."""
    from nlt_amd.networks.elements import Conv2D
    rng = np.random.default_rng(c + h)
    n, cin = 2, 2 * c
    R = lambda *shape: rng.standard_normal(shape).astype(np.float32)
    wk = R(2, 2, cin, cout)
    layer = Conv2D(cout, 2, 2)
    layer.set_weights(wk, np.zeros(cout, np.float32))
    oh, ow = h // 2, w // 2
    dp, existing, fm_y, obs_y, dobs0 = R(n, oh, ow, cout), R(n, h, w, cin), R(n, h, w, cin), R(n, h, w, c), R(n, h, w, c)
    zb = torch.zeros(256, device='cuda')
    outs = []
    for tiled in (False, True):
        out, dobs = d(existing), d(dobs0)
        kw = dict(mask_src=d(fm_y), ldm=cin, mask_alpha=0.3, accumulate=True, split=(c, d(obs_y), dobs, 0.2, partial))
        if tiled:
            C.conv_tile_backward_data(C.DECONV_K2S2, d(dp), cout, cout, n, oh, ow, layer.packed_adjoint_tile(0, cin, tn), cin, tn, out, cin, **kw)
        else:
            C.conv_backward_data(C.DECONV_K2S2, d(dp), cout, cout, n, oh, ow, layer.packed_adjoint(0, cin)[0], zb, cin, out, cin, **kw)
        torch.cuda.synchronize()
        outs.append((out.cpu().numpy(), dobs.cpu().numpy()))
    tol = 3e-5 * float(np.abs(outs[0][0]).max())
    np.testing.assert_allclose(outs[1][0], outs[0][0], atol=tol)
    np.testing.assert_allclose(outs[1][1], outs[0][1], atol=tol)
    np.testing.assert_array_equal(outs[1][0][..., c:], existing[..., c:])


@pytest.mark.parametrize('mode,ksplit,partial', [(C.CONV_K2S2, 1, True), (C.CONV_K2S2, 4, True), (C.DECONV_K2S2, 1, False),
                                                 (C.DECONV_K2S2, 2, True), (C.CONV_K2S1, 1, True)])
def test_backward_data_with_the_level_split_epilogue(mode, ksplit, partial):
    """nlt_conv_backward_data: backward-data of a conv whose input is fm[l] = [query c | observation-mean c] (one observation
    per frame), accumulated on top of what dfm[l] already holds, with the `level_split` work in its epilogue -- query half:
    times LeakyReLU'(fm y) into dfm[l]; observation half: (+ the observation path's own gradient) times LeakyReLU'(obs y)
    into dobs -- against the separate steps (torch autograd for the conv's input gradient, the formulas of
    nlt_level_split_backward); single-pass and split-K."""
    from nlt_amd.networks.elements import Conv2D
    rng = np.random.default_rng(mode * 10 + ksplit)
    k, s, tr = MODES[mode]
    c, cout, n, h, w = 16, 48, 2, 8, 12
    cin = 2 * c
    R = lambda *shape: rng.standard_normal(shape).astype(np.float32)
    wk = R(*((2, 2, cout, cin) if tr else (2, 2, cin, cout)))
    x = torch.tensor(R(n, h, w, cin), requires_grad=True)
    y = (T.conv2d_transpose_same if tr else T.conv2d_same)(x, torch.tensor(wk), torch.zeros(cout), s)
    dp = R(*y.shape)
    (gx,) = torch.autograd.grad(y, x, torch.tensor(dp))
    existing, fm_y, obs_y, dobs0 = R(n, h, w, cin), R(n, h, w, cin), R(n, h, w, c), R(n, h, w, c)
    aq, ao = 0.3, 0.2
    tot = gx + torch.tensor(existing)
    ref_q = tot[..., :c] * torch.where(torch.tensor(fm_y[..., :c]) > 0, 1.0, aq)
    ref_o = (tot[..., c:] + (torch.tensor(dobs0) if partial else 0)) * torch.where(torch.tensor(obs_y) > 0, 1.0, ao)
    layer = Conv2D(cout, 2, s, transpose=tr)
    layer.set_weights(wk, np.zeros(cout, np.float32))
    packed, ks = layer.packed_adjoint(0, cin)
    oh, ow = y.shape[1:3]
    out, dobs = d(existing), d(dobs0)
    C.conv_backward_data(layer.ADJOINT[mode], d(dp), cout, cout, n, oh, ow, packed, torch.zeros(64, device='cuda'), cin, out, cin,
                         mask_src=d(fm_y), ldm=cin, mask_alpha=aq, accumulate=True, ksplit=ksplit,
                         split=(c, d(obs_y), dobs, ao, partial))
    torch.cuda.synchronize()
    tol = 3e-5 * float(tot.abs().max())
    np.testing.assert_allclose(out.cpu().numpy()[..., :c], ref_q.numpy(), atol=tol)
    np.testing.assert_allclose(dobs.cpu().numpy(), ref_o.numpy(), atol=tol)
    np.testing.assert_array_equal(out.cpu().numpy()[..., c:], existing[..., c:])     # the observation half of dfm[l] is not rewritten


def test_lrelu_and_obs_mean_backward():
    rng = np.random.default_rng(0)
    n, k, hw, c = 2, 3, 35, 16
    g = rng.standard_normal((n * hw, 2 * c)).astype(np.float32)
    y = rng.standard_normal((n * hw, 2 * c)).astype(np.float32)
    gd = d(g)
    C.lrelu_backward(gd, 2 * c, d(y), 2 * c, c, n * hw, 0.3, gd, 2 * c)           # in place on the first half
    ref = g.copy(); ref[:, :c] = g[:, :c] * np.where(y[:, :c] > 0, 1.0, 0.3)
    np.testing.assert_allclose(gd.cpu().numpy(), ref, atol=1e-6)
    dmean = rng.standard_normal((n, hw, 2 * c)).astype(np.float32)
    obs_y = rng.standard_normal((n, k, hw, c)).astype(np.float32)
    part = rng.standard_normal((n, k, hw, c)).astype(np.float32)
    ow = rng.random((n, k), dtype=np.float32)
    for use_w, use_p, use_y in ((True, True, True), (False, False, True), (False, True, False)):
        pd = d(part)
        C.obs_mean_backward(d(dmean).view(-1)[c:], 2 * c, d(obs_y) if use_y else None, d(ow) if use_w else None,
                            pd if use_p else None, n, k, hw, c, 0.3, pd)
        ref = np.broadcast_to(dmean[:, None, :, c:] / k, (n, k, hw, c)).copy()
        if use_w: ref = ref * ow[:, :, None, None]
        if use_p: ref = ref + part
        if use_y: ref = ref * np.where(obs_y > 0, 1.0, 0.3)
        np.testing.assert_allclose(pd.cpu().numpy(), ref, atol=1e-6)


def test_l2_train_loss_equals_its_parts():
    """nlt_l2_train_loss = nlt_mul_forward + nlt_l2_loss_forward (+ sum / global batch) + nlt_l2_loss_backward in one launch."""
    rng = np.random.default_rng(3)
    n, h, w, gbs = 3, 37, 29, 8
    pred, rgb = rng.random((n, h, w, 3), dtype=np.float32), rng.random((n, h, w, 3), dtype=np.float32)
    fg = (rng.random((n, h, w, 1)) > 0.3).astype(np.float32).repeat(3, -1)
    loss, gt, dpred = C.l2_train_loss(d(pred), d(rgb), d(fg), gbs)
    gt_ref = C.mul_forward(d(rgb), d(fg))
    per = C.l2_loss_forward(d(pred), gt_ref)
    gl = torch.full((n,), 1.0 / gbs, device='cuda')
    assert torch.equal(gt, gt_ref)
    np.testing.assert_allclose(float(loss), float(per.sum()) / gbs, rtol=1e-6)
    np.testing.assert_allclose(dpred.cpu().numpy(), C.l2_loss_backward(d(pred), gt_ref, gl).cpu().numpy(), rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize('k,use_w,use_p', [(1, False, True), (3, True, True), (4, False, False)])
def test_level_split_backward_equals_the_two_launches(k, use_w, use_p):
    """nlt_level_split_backward = nlt_lrelu_backward on the query half + nlt_obs_mean_backward on the observation half,
    bit for bit (same operations per element), in one launch."""
    rng = np.random.default_rng(k)
    n, hw, c = 2, 77, 32
    dfm = rng.standard_normal((n, hw, 2 * c)).astype(np.float32)
    fm_y = rng.standard_normal((n, hw, 2 * c)).astype(np.float32)
    obs_y = rng.standard_normal((n, k, hw, c)).astype(np.float32)
    part = rng.standard_normal((n, k, hw, c)).astype(np.float32)
    ow = rng.random((n, k), dtype=np.float32)
    g2, out2 = d(dfm), d(part)
    C.obs_mean_backward(g2.view(-1)[c:], 2 * c, d(obs_y), d(ow) if use_w else None, out2 if use_p else None, n, k, hw, c, 0.2, out2)
    C.lrelu_backward(g2, 2 * c, d(fm_y), 2 * c, c, n * hw, 0.3, g2, 2 * c)
    g1, out1 = d(dfm), d(part)
    C.level_split_backward(g1, d(fm_y), 2 * c, d(obs_y), d(ow) if use_w else None, out1 if use_p else None, n, k, hw, c, 0.3, 0.2, out1)
    assert torch.equal(g1, g2) and torch.equal(out1, out2)


@pytest.mark.parametrize('k,weights,partial', [(1, False, True), (3, True, True), (2, False, False)])
def test_stem_backward(k, weights, partial):
    rng = np.random.default_rng(k)
    n, h, w, c = 2, 9, 7, 16
    U = lambda *s: rng.random(s, dtype=np.float32)
    base, cvis, lvis, nn_rgb, nn_base = U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1), U(n, k, h, w, 3), U(n, k, h, w, 3)
    ow = U(n, k) if weights else None
    dfm0 = rng.standard_normal((n, h, w, 2 * c)).astype(np.float32)
    dobs0 = rng.standard_normal((n, k, h, w, c)).astype(np.float32) if partial else None
    Z = lambda *s: torch.zeros(s, device='cuda')
    dwq, dbq, dwo, dbo = Z(1, 1, 5, c), Z(c), Z(1, 1, 3, c), Z(c)
    C.stem_backward(d(base), d(cvis), d(lvis), d(nn_rgb), d(nn_base), d(ow), n, k, h, w, c, d(dfm0), d(dobs0), dwq, dbq, dwo, dbo)
    x = np.concatenate((base, cvis, lvis), -1).reshape(-1, 5).astype(np.float64)
    gq = dfm0[..., :c].reshape(-1, c).astype(np.float64)
    gm = np.broadcast_to(dfm0[:, None, ..., c:] / k, (n, k, h, w, c)).astype(np.float64)
    if weights: gm = gm * ow[:, :, None, None, None]
    g = gm + (dobs0 if partial else 0)
    dd = (nn_rgb - nn_base).reshape(-1, 3).astype(np.float64)
    np.testing.assert_allclose(dwq.cpu().numpy()[0, 0], x.T @ gq, atol=2e-4)
    np.testing.assert_allclose(dbq.cpu().numpy(), gq.sum(0), atol=2e-4)
    np.testing.assert_allclose(dwo.cpu().numpy()[0, 0], dd.T @ g.reshape(-1, c), atol=2e-4)
    np.testing.assert_allclose(dbo.cpu().numpy(), g.reshape(-1, c).sum(0), atol=2e-4)


def test_head_backward():
    rng = np.random.default_rng(1)
    n, h, w, cd, cs = 2, 7, 9, 4, 32
    dec = rng.standard_normal((n, h, w, cd)).astype(np.float32); skip = rng.standard_normal((n, h, w, cs)).astype(np.float32)
    wk = rng.standard_normal((1, 1, cd + cs, 3)).astype(np.float32)
    dpred = rng.standard_normal((n, h, w, 3)).astype(np.float32)
    d_dec = torch.empty(n, h, w, cd, device='cuda'); d_skip = torch.empty(n, h, w, cs, device='cuda')
    dw = torch.zeros(1, 1, cd + cs, 3, device='cuda'); db = torch.zeros(3, device='cuda')
    C.head_backward(d(dec), cd, cd, d(skip), cs, cs, d(wk), d(dpred), n, h, w, d_dec, cd, d_skip, cs, dw, db)
    g = dpred.copy(); g[:, 0, 0, :] = 0
    dx = g @ wk[0, 0].T
    np.testing.assert_allclose(d_dec.cpu().numpy(), dx[..., :cd], atol=1e-5)
    np.testing.assert_allclose(d_skip.cpu().numpy(), dx[..., cd:], atol=1e-5)
    x = np.concatenate((dec, skip), -1).reshape(-1, cd + cs).astype(np.float64)
    np.testing.assert_allclose(dw.cpu().numpy()[0, 0], x.T @ g.reshape(-1, 3), atol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), g.reshape(-1, 3).sum(0), atol=2e-4)


def test_warp_and_resize_backward():
    rng = np.random.default_rng(2)
    n, uvh, uvw, hc, wc = 2, 16, 24, 12, 10
    warp = rng.random((n, hc, wc, 2), dtype=np.float32).astype(np.float16).astype(np.float32)
    warp[rng.random((n, hc, wc)) > 0.7] = 0
    warp[0, 0, 0] = (1.0, 0.2); warp[0, 0, 1] = ((uvw - 1) / uvw, (uvh - 1) / uvh)
    dcam = rng.standard_normal((n, hc, wc, 3)).astype(np.float32)
    dpred = torch.empty(n, uvh, uvw, 3, device='cuda')
    C.warp_backward(d(dcam), d(warp), n, uvh, uvw, hc, wc, dpred)
    data = torch.zeros(n, uvh, uvw, 3, requires_grad=True)
    out = T.resampler(T.set_left_top_corner(data, 0), torch.tensor(warp) * torch.tensor([uvw, uvh], dtype=torch.float32))
    (ref,) = torch.autograd.grad(out, data, torch.tensor(dcam))
    np.testing.assert_allclose(dpred.cpu().numpy(), ref.numpy(), atol=1e-5)
    assert np.all(dpred.cpu().numpy()[:, 0, 0, :] == 0)
    dout = rng.standard_normal((2, 20, 14, 3)).astype(np.float32)
    x = torch.zeros(2, 12, 10, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(T.resize_bilinear(x, 20, 14), x, torch.tensor(dout))
    np.testing.assert_allclose(C.resize_bilinear_backward(d(dout), 12, 10).cpu().numpy(), ref.numpy(), atol=1e-5)


def test_l2_loss_and_scale_rows():
    rng = np.random.default_rng(3)
    pred, gt = rng.random((3, 17, 19, 3), dtype=np.float32), rng.random((3, 17, 19, 3), dtype=np.float32)
    ref = O.l2_loss(torch.tensor(gt), torch.tensor(pred), keep_batch=True).numpy()
    np.testing.assert_allclose(C.l2_loss_forward(d(pred), d(gt)).cpu().numpy(), ref, rtol=2e-6)
    gl = rng.random(3, dtype=np.float32)
    got = C.l2_loss_backward(d(pred), d(gt), d(gl)).cpu().numpy()
    np.testing.assert_allclose(got, gl[:, None, None, None] * 2 * (pred - gt) / (17 * 19 * 3), atol=1e-7)
    np.testing.assert_allclose(C.scale_rows(d(pred), d(gl)).cpu().numpy(), pred * gl[:, None, None, None], atol=1e-7)


@pytest.mark.parametrize('wshape', [(3, 17, 19), (3, 17, 19, 1), (3, 1, 1), ()])
@pytest.mark.parametrize('keep_batch', [True, False])
def test_l2_loss_with_sample_weights(wshape, keep_batch):
    """losses.L2(weights=) = Keras `sample_weight` on the [N,H,W] loss map (nlt/losses.py:42-43): value and gradient
    w.r.t. pred against the oracle's float64 autograd, for every weight shape Keras broadcasts."""
    from nlt_amd import losses
    rng = np.random.default_rng(31)
    pred = torch.tensor(rng.random((3, 17, 19, 3), dtype=np.float32), requires_grad=True)
    gt = torch.tensor(rng.random((3, 17, 19, 3), dtype=np.float32))
    wt = torch.tensor(rng.random(wshape, dtype=np.float32) if wshape else np.float32(0.625))
    ref = O.l2_loss(gt.double(), pred.double(), keep_batch, weights=wt.double())
    (gref,) = torch.autograd.grad(ref.sum(), pred)
    pg = pred.detach().cuda().requires_grad_(True)
    got = losses.L2()(gt.cuda(), pg, keep_batch=keep_batch, weights=wt.cuda())
    (ggot,) = torch.autograd.grad(got.sum(), pg)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-6)
    assert float((ggot.cpu() - gref).norm() / gref.norm()) < 1e-6


def test_barron_loss_with_alpha_blend_weights():
    """losses.Barron(weights=): gt and pred alpha-blended against zeros before the loss (nlt/losses.py:107-110)."""
    from nlt_amd import losses
    rng = np.random.default_rng(32)
    h, w = 32, 48
    pred = torch.tensor(rng.random((2, h, w, 3), dtype=np.float32), requires_grad=True)
    gt = torch.tensor(rng.random((2, h, w, 3), dtype=np.float32))
    alpha = torch.tensor((rng.random((2, h, w, 1)) > 0.3).astype(np.float32) * rng.random((2, h, w, 1), dtype=np.float32))
    ref = B.barron_loss(gt.double(), pred.double(), keep_batch=True, weights=alpha.double())
    (gref,) = torch.autograd.grad(ref.sum(), pred)
    pg = pred.detach().cuda().requires_grad_(True)
    got = losses.Barron(w, h)(gt.cuda(), pg, keep_batch=True, weights=alpha.cuda())
    (ggot,) = torch.autograd.grad(got.sum(), pg)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5)
    assert float((ggot.cpu() - gref).norm() / gref.norm()) < 1e-4


@pytest.mark.parametrize('h,w', [(64, 64), (32, 48), (83, 71), (17, 17)])
def test_barron_loss_and_grad(h, w):
    rng = np.random.default_rng(h)
    pred = torch.tensor(rng.random((2, h, w, 3), dtype=np.float32), requires_grad=True)
    gt = torch.tensor(rng.random((2, h, w, 3), dtype=np.float32))
    ref = B.barron_loss(gt.double(), pred.double(), keep_batch=True)
    (gref,) = torch.autograd.grad(ref.sum(), pred)
    loss, dunit = C.barron_loss(pred.detach().cuda(), gt.cuda(), True)
    np.testing.assert_allclose(loss.cpu().numpy(), ref.detach().numpy(), rtol=2e-5)
    assert float((dunit.cpu() - gref).norm() / gref.norm()) < 1e-4
    loss2, none = C.barron_loss(pred.detach().cuda(), gt.cuda(), False)
    assert none is None
    np.testing.assert_allclose(loss2.cpu().numpy(), loss.cpu().numpy(), rtol=1e-6)


def test_barron_wavelet_matches_reference_golden():
    """The kernel's CDF9/7 pyramid reproduces the reference's wavelet_golden.mat through the loss:
    loss(x) for x = golden image equals the Charbonnier NLL of the GOLDEN coefficients."""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'wavelet_golden.npz'))
    im = g['I_color'].astype(np.float64)                       # (3, 83, 71): already "sYUV planes"
    coefs = np.concatenate([g[k].reshape(-1) for k in g.files if k.startswith('band_') or k == 'resid'])
    nll = np.sqrt((coefs / 0.01) ** 2 + 1) - 1 + np.log(0.01) + B.LOG_Z_ALPHA1
    # feed residual r with syuv(r) == im:  r = im^T @ inv(M)
    M = B.RGB_TO_YUV * B.VOLUME_PRESERVING_YUV_SCALE
    r = np.transpose(im, (1, 2, 0)) @ np.linalg.inv(M)
    gt = torch.tensor(r[None].astype(np.float32)); pred = torch.zeros_like(gt)
    loss, _ = C.barron_loss(pred.cuda(), gt.cuda(), False)
    assert abs(loss.item() - nll.mean()) < 2e-3 * abs(nll.mean())   # fp32 input rounding through /0.01


def test_adam_amsgrad_matches_keras_form():
    rng = np.random.default_rng(4)
    p0 = rng.standard_normal(1001).astype(np.float32)
    p = torch.tensor(p0.copy(), requires_grad=True)
    opt = O.KerasAdamAMSGrad([p], 1e-2)
    pd = d(p0.copy()); m = torch.zeros_like(pd); v = torch.zeros_like(pd); vh = torch.zeros_like(pd)
    import math
    for t in range(1, 4):
        g = rng.standard_normal(1001).astype(np.float32) * (1.0 if t != 2 else 0.1)   # t=2: vhat keeps the max
        opt.step([torch.tensor(g)])
        lr_t = 1e-2 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        C.adam_amsgrad_step(pd, d(g), m, v, vh, lr_t, 0.9, 0.999, 1e-7)
        np.testing.assert_allclose(pd.cpu().numpy(), p.detach().numpy(), atol=2e-6)


def test_clip_by_norm_slots_matches_tf_clip_by_norm():
    """Keras clipnorm (mgm > 0): per-slot t * clip / max(||t||, clip) over the flat gradient bucket, in place; slots that are
    below the norm come back multiplied-then-divided (tf.clip_by_norm's op order), an all-zero slot stays zero."""
    from oracle import nlt_oracle as O
    g = torch.Generator(device='cuda').manual_seed(5)
    sizes = [16, 5 * 16, 2 * 2 * 32 * 16, 3, 1024 * 1024 + 7, 64, 8]
    offs, off = [], 0
    for n in sizes:
        offs.append((off, n)); off += (n + 3) // 4 * 4
    flat = torch.randn(off, device='cuda', generator=g) * 1e-3
    flat[offs[5][0]:offs[5][0] + 64] = 0                         # an all-zero tensor
    flat[offs[2][0]:offs[2][0] + sizes[2]] *= 50                 # one far above the threshold
    ref = flat.clone().cpu()
    clip = 0.02
    for o, n in offs:
        ref[o:o + n] = O.clip_by_norm(ref[o:o + n].clone(), clip)
    slots = torch.tensor(offs, dtype=torch.int64, device='cuda')
    C.clip_by_norm_slots(flat, slots, clip)
    torch.cuda.synchronize()
    got = flat.cpu()
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    for o, n in offs:
        assert float(got[o:o + n].norm()) <= clip * (1 + 1e-5)
    pad = torch.ones(off, dtype=torch.bool)
    for o, n in offs:
        pad[o:o + n] = False
    assert torch.equal(got[pad], ref[pad])                       # padding between slots untouched
