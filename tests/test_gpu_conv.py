"""-m gpu: every conv family of libnlt_hip.so (both algorithms, several wave tiles, ragged row
counts, dual-source virtual concat, channel-slice outputs, backward-data epilogue) against the
CPU oracle's TF-semantics primitives.  fp32 tolerance: 2e-5 relative to the output scale
(K <= 2048 fp32 accumulation, different summation order)."""
import os

import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

MODES = {C.CONV1X1: (1, 1, False), C.CONV_K2S2: (2, 2, False), C.CONV_K2S1: (2, 1, False),
         C.DECONV_K2S2: (2, 2, True), C.DECONV_K2S1: (2, 1, True)}


def oracle_conv(mode, x, wk, b, act, alpha):
    k, s, tr = MODES[mode]
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    y = f(torch.tensor(x), torch.tensor(wk), torch.tensor(b), s)
    if act:
        y = T.leaky_relu(y, alpha)
    return y.numpy()


def run_case(mode, n, h, w, c0, c1, cout, algo, tile_hint=0, act=True, alpha=0.3, pad0=0, pad1=0, pado=0, seed=0, ksplit=1):
    rng = np.random.default_rng(seed)
    k, s, tr = MODES[mode]
    cin = c0 + c1
    x0 = rng.standard_normal((n, h, w, c0 + pad0)).astype(np.float32)
    x1 = rng.standard_normal((n, h, w, c1 + pad1)).astype(np.float32) if c1 else None
    wk = (rng.standard_normal((k, k, cout, cin) if tr else (k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    x = x0[..., :c0] if not c1 else np.concatenate((x0[..., :c0], x1[..., :c1]), -1)
    ref = oracle_conv(mode, x, wk, b, act, alpha)
    oh, ow = ref.shape[1:3]
    d = lambda a: None if a is None else torch.tensor(a).cuda()
    out = torch.full((n, oh, ow, cout + pado), -7.0, device='cuda')
    wd = d(wk)
    packed = C.pack_conv_weights(mode, wd, c0, c1, cout) if algo != C.ALGO_DIRECT else None
    if ksplit > 1:
        C.conv_forward_splitk(mode, ksplit, d(x0), c0, c0 + pad0, d(x1), c1, c1 + pad1, n, h, w, packed, d(b), cout, out,
                              cout + pado, act=act, alpha=alpha, tile_hint=tile_hint)
    else:
        C.conv_forward(mode, d(x0), c0, c0 + pad0, d(x1), c1, c1 + pad1, n, h, w, wd, packed, d(b), cout, out,
                       cout + pado, act=act, alpha=alpha, algo=algo, tile_hint=tile_hint)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    scale = max(np.abs(ref).max(), 1.0)
    np.testing.assert_allclose(got[..., :cout], ref, atol=2e-5 * scale, rtol=0)
    if pado:
        assert np.all(got[..., cout:] == -7.0), "wrote outside its channel slice"


@pytest.mark.parametrize('mode', list(MODES))
@pytest.mark.parametrize('algo', [C.ALGO_DIRECT, C.ALGO_MFMA])
def test_conv_modes_basic(mode, algo):
    run_case(mode, 2, 8, 12, 16, 0, 32, algo)


@pytest.mark.parametrize('mode', list(MODES))
@pytest.mark.parametrize('tile', [0x11, 0x12, 0x14, 0x21, 0x22, 0x24, 0x41, 0x42, 0x44])
def test_conv_mfma_tiles_ragged(mode, tile):
    # 3*6*10 = 180 rows (not a multiple of 16), 64 outputs so every CT divides
    run_case(mode, 3, 6, 10, 32, 0, 64, C.ALGO_MFMA, tile_hint=tile, seed=tile)


@pytest.mark.parametrize('mode', list(MODES))
@pytest.mark.parametrize('algo', [C.ALGO_DIRECT, C.ALGO_MFMA])
def test_conv_dual_source_and_slices(mode, algo):
    # virtual concat of two sources that are channel slices of wider tensors; sliced output
    run_case(mode, 1, 6, 4, 16, 48, 16, algo, pad0=16, pad1=8, pado=16, seed=3)


@pytest.mark.parametrize('mode,c0,c1,cout', [
    (C.DECONV_K2S2, 8, 32, 4),      # L12a: K = 8 | 32 (padded chunks), N = 4*4
    (C.DECONV_K2S1, 4, 0, 4),       # L12b
    (C.DECONV_K2S1, 8, 0, 8),       # L11b
    (C.CONV1X1, 4, 32, 12),         # head-like, N not a multiple of 16
    (C.CONV_K2S2, 20, 12, 24)])     # odd multiples of 4 everywhere
def test_conv_mfma_small_channels(mode, c0, c1, cout):
    run_case(mode, 2, 6, 6, c0, c1, cout, C.ALGO_MFMA, seed=5)


@pytest.mark.parametrize('mode,c0,c1,cout', [(C.CONV1X1, 5, 0, 16), (C.CONV1X1, 4, 32, 3), (C.CONV_K2S1, 3, 2, 7),
                                             (C.DECONV_K2S2, 5, 3, 3), (C.DECONV_K2S1, 7, 0, 5), (C.CONV_K2S2, 6, 0, 9)])
def test_conv_direct_any_channels(mode, c0, c1, cout):
    run_case(mode, 2, 4, 6, c0, c1, cout, C.ALGO_DIRECT, seed=7)


@pytest.mark.parametrize('algo', [C.ALGO_DIRECT, C.ALGO_MFMA])
def test_conv_no_act_and_relu(algo):
    run_case(C.CONV_K2S1, 1, 5, 5, 16, 0, 16, algo, act=False)
    run_case(C.CONV_K2S1, 1, 5, 5, 16, 0, 16, algo, act=True, alpha=0.0)


@pytest.mark.parametrize('algo', [C.ALGO_DIRECT, C.ALGO_MFMA])
def test_conv_deep_k2048(algo):
    # L6-like: K = 4*512, 256 outputs, tiny spatial extent (1x1 output grid per frame)
    run_case(C.CONV_K2S2, 3, 2, 2, 512, 0, 256, algo, seed=11)
    run_case(C.DECONV_K2S2, 2, 1, 1, 512, 512, 128, algo, seed=12)


@pytest.mark.parametrize('mode', list(MODES))
@pytest.mark.parametrize('ksplit,tile', [(2, 0x11), (3, 0x12), (8, 0x22), (64, 0x11)])
def test_conv_split_k(mode, ksplit, tile):
    # K slices that do not divide the chunk count, more slices than chunks (clamped), dual source, sliced output
    run_case(mode, 2, 6, 10, 48, 16, 32, C.ALGO_MFMA, tile_hint=tile, pad0=8, pado=16, seed=ksplit, ksplit=ksplit)


def test_conv_split_k_deep_levels():
    run_case(C.CONV_K2S2, 4, 16, 16, 512, 0, 256, C.ALGO_MFMA, tile_hint=0x12, seed=21, ksplit=8)       # L6.q.s2 shape
    run_case(C.DECONV_K2S2, 4, 8, 8, 512, 512, 128, C.ALGO_MFMA, tile_hint=0x12, seed=22, ksplit=8)     # L7.q.s2 shape
    run_case(C.DECONV_K2S1, 4, 16, 16, 128, 0, 128, C.ALGO_MFMA, tile_hint=0x11, seed=23, ksplit=4)     # L7.q.s1 shape


@pytest.mark.parametrize('mode,n,h,w,cin,cout,tile,ksplit', [
    (C.CONV_K2S1, 4, 1, 1, 1024, 1024, 0x12, 64),       # depth-1024 bottleneck: 4 GEMM rows, one live tap
    (C.CONV_K2S2, 4, 2, 2, 2048, 1024, 0x14, 128),      # 4 rows, K = 8192
    (C.DECONV_K2S2, 4, 2, 2, 1024, 512, 0x12, 16),
    (C.DECONV_K2S1, 4, 4, 4, 512, 512, 0x22, 32),
    (C.CONV_K2S2, 4, 32, 32, 128, 256, 0x22, 8),        # mid-network: 1024 rows, two groups of four slices
    (C.CONV_K2S1, 2, 16, 16, 256, 256, 0x44, 6),        # the widest wave tile (48 KB of LDS for the meeting)
    (C.CONV_K2S2, 4, 2, 2, 2048, 1024, 0x12, -128),     # the two-launch form (ksplit < 0): slices by independent waves + a summing launch
    (C.DECONV_K2S2, 4, 2, 2, 1024, 512, 0x12, -16),
    (C.CONV_K2S2, 4, 32, 32, 128, 256, 0x22, -8),
])
def test_conv_split_k_in_launch_reduction_is_reproducible_and_fresh(mode, n, h, w, cin, cout, tile, ksplit):
    """The in-launch split-K reduction (csrc/conv_mfma.hip: groups of four slices meet in LDS, groups meet through the workspace,
    last-arriving workgroup finishes the tile): the SAME workspace serves launch after launch with new inputs while another stream
    keeps the memory system busy -- a reducer that read stale partial tiles (per-XCD L2s, per-CU L1) or a counter left non-zero
    would show here.  Each result against the single-slice launch (<= 2e-5 of the output scale); a repeated launch bit-identical."""
    g = torch.Generator(device='cuda').manual_seed(abs(ksplit) * 7 + cin)
    k, s, tr = MODES[mode]
    wk = torch.randn((k, k, cout, cin) if tr else (k, k, cin, cout), device='cuda', generator=g) / (k * k * cin) ** 0.5
    packed = C.pack_conv_weights(mode, wk, cin, 0, cout)
    bias = torch.randn(cout, device='cuda', generator=g)
    oh, ow = (h // 2, w // 2) if mode == C.CONV_K2S2 else ((2 * h, 2 * w) if mode == C.DECONV_K2S2 else (h, w))
    out, ref, again = (torch.empty(n, oh, ow, cout, device='cuda') for _ in range(3))
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device='cuda')
    for it in range(int(os.environ.get('NLT_STRESS_ITERS', '12'))):
        x = torch.randn(n, h, w, cin, device='cuda', generator=g) * (1.0 + it % 7)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):                     # streaming traffic on every XCD while the split launches run
            for _ in range(4):
                big.add_(1.0)
        C.conv_forward_splitk(mode, ksplit, x, cin, cin, None, 0, 0, n, h, w, packed, bias, cout, out, cout, tile_hint=tile)
        C.conv_forward_splitk(mode, ksplit, x, cin, cin, None, 0, 0, n, h, w, packed, bias, cout, again, cout, tile_hint=tile)
        C.conv_forward(mode, x, cin, cin, None, 0, 0, n, h, w, wk, packed, bias, cout, ref, cout, algo=C.ALGO_MFMA, tile_hint=tile)
        torch.cuda.synchronize()
        scale = max(float(ref.abs().max()), 1.0)
        assert float((out - ref).abs().max()) <= 2e-5 * scale, (it, float((out - ref).abs().max()), scale)
        assert torch.equal(out, again), it


def test_conv_split_k_launch_shapes_alternating_on_one_workspace():
    """Launches of different shapes, slice counts and forms (one launch / two launches) share ONE workspace per stream: the ticket
    counters of one launch are the partial-tile region of none, and every launch leaves its counters at zero -- 300 launches in random
    order, each checked against the single-slice launch."""
    g = torch.Generator(device='cuda').manual_seed(77)
    cases = []
    for mode, n, h, w, cin, cout, tile, ks in ((C.CONV_K2S1, 4, 1, 1, 1024, 1024, 0x12, 64), (C.CONV_K2S2, 4, 32, 32, 128, 256, 0x22, 8),
                                                (C.DECONV_K2S2, 4, 2, 2, 1024, 512, 0x12, 16), (C.CONV_K2S2, 4, 2, 2, 2048, 1024, 0x12, -128),
                                                (C.DECONV_K2S1, 4, 4, 4, 512, 512, 0x22, 32), (C.CONV_K2S1, 2, 16, 16, 256, 256, 0x44, 6),
                                                (C.CONV_K2S2, 4, 64, 64, 64, 128, 0x22, 4), (C.CONV_K2S2, 4, 32, 32, 128, 256, 0x11, -8)):
        k, s, tr = MODES[mode]
        wk = torch.randn((k, k, cout, cin) if tr else (k, k, cin, cout), device='cuda', generator=g) / (k * k * cin) ** 0.5
        oh, ow = (h // 2, w // 2) if mode == C.CONV_K2S2 else ((2 * h, 2 * w) if mode == C.DECONV_K2S2 else (h, w))
        cases.append(dict(mode=mode, n=n, h=h, w=w, cin=cin, cout=cout, tile=tile, ks=ks, wk=wk, packed=C.pack_conv_weights(mode, wk, cin, 0, cout),
                          bias=torch.randn(cout, device='cuda', generator=g), out=torch.empty(n, oh, ow, cout, device='cuda'),
                          ref=torch.empty(n, oh, ow, cout, device='cuda')))
    order = torch.randint(0, len(cases), (int(os.environ.get('NLT_STRESS_ITERS', '300')),), generator=torch.Generator().manual_seed(5)).tolist()
    for it, ci in enumerate(order):
        c = cases[ci]
        x = torch.randn(c['n'], c['h'], c['w'], c['cin'], device='cuda', generator=g)
        C.conv_forward_splitk(c['mode'], c['ks'], x, c['cin'], c['cin'], None, 0, 0, c['n'], c['h'], c['w'], c['packed'], c['bias'], c['cout'],
                              c['out'], c['cout'], tile_hint=c['tile'])
        C.conv_forward(c['mode'], x, c['cin'], c['cin'], None, 0, 0, c['n'], c['h'], c['w'], c['wk'], c['packed'], c['bias'], c['cout'],
                       c['ref'], c['cout'], algo=C.ALGO_MFMA, tile_hint=c['tile'])
        if it % 25 == 24 or it == len(order) - 1:
            torch.cuda.synchronize()
        err = (c['out'] - c['ref']).abs().max()
        scale = c['ref'].abs().max().clamp(min=1.0)
        assert float(err) <= 2e-5 * float(scale), (it, ci, float(err), float(scale))


@pytest.mark.parametrize('algo', [C.ALGO_DIRECT, C.ALGO_MFMA])
def test_conv_backward_data_epilogue(algo):
    """mask_src / accumulate: out = (old + conv(x)) * lrelu'(mask)."""
    rng = np.random.default_rng(1)
    n, h, w, cin, cout = 2, 4, 6, 16, 16
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wk = rng.standard_normal((2, 2, cin, cout)).astype(np.float32) * 0.1
    b = np.zeros(cout, np.float32)
    old = rng.standard_normal((n, h, w, cout)).astype(np.float32)
    mask = rng.standard_normal((n, h, w, cout)).astype(np.float32)
    ref = (old + oracle_conv(C.CONV_K2S1, x, wk, b, False, 0)) * np.where(mask > 0, 1.0, 0.3)
    d = lambda a: torch.tensor(a).cuda()
    out = d(old)
    wd = d(wk)
    packed = C.pack_conv_weights(C.CONV_K2S1, wd, cin, 0, cout)
    C.conv_forward(C.CONV_K2S1, d(x), cin, cin, None, 0, 0, n, h, w, wd, packed, d(b), cout, out, cout,
                   act=False, alpha=0.3, algo=algo, mask_src=d(mask), ldm=cout, accumulate=True)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5 * np.abs(ref).max())


def test_conv_rejects_bad_args():
    x = torch.zeros(1, 3, 3, 16, device='cuda')
    w = torch.zeros(2, 2, 16, 16, device='cuda')
    b = torch.zeros(16, device='cuda')
    out = torch.zeros(1, 3, 3, 16, device='cuda')
    with pytest.raises(C.NLTError):     # odd size for k2s2 (TF would pad; reference never does this)
        C.conv_forward(C.CONV_K2S2, x, 16, 16, None, 0, 0, 1, 3, 3, w, None, b, 16, out, 16, algo=C.ALGO_DIRECT)
    with pytest.raises(C.NLTError):     # MFMA requested without packed weights
        C.conv_forward(C.CONV_K2S1, x, 16, 16, None, 0, 0, 1, 3, 3, w, None, b, 16, out, 16, algo=C.ALGO_MFMA)
    with pytest.raises(C.NLTError):     # CPU tensor
        C.conv_forward(C.CONV_K2S1, x.cpu(), 16, 16, None, 0, 0, 1, 3, 3, w, None, b, 16, out, 16)
