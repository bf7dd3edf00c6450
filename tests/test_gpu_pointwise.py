"""-m gpu: stem / observation mean / head / warp (bit-exact integer UV indices) / resize / mul
against the CPU oracle."""
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu
d = lambda a: None if a is None else torch.tensor(a).cuda()


@pytest.mark.parametrize('k,weights', [(1, False), (3, False), (4, True)])
def test_stem(k, weights):
    rng = np.random.default_rng(k)
    n, h, w, c = 2, 5, 7, 16
    U = lambda *s: rng.random(s, dtype=np.float32)
    base, cvis, lvis = U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1)
    nn_rgb, nn_base = U(n, k, h, w, 3), U(n, k, h, w, 3)
    wq, bq = rng.standard_normal((1, 1, 5, c)).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    wo, bo = rng.standard_normal((1, 1, 3, c)).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    ow = U(n, k) if weights else None
    fm0 = torch.empty(n, h, w, 2 * c, device='cuda'); obs0 = torch.empty(n, k, h, w, c, device='cuda')
    C.stem_forward(d(base), d(cvis), d(lvis), d(nn_rgb), d(nn_base), d(ow), n, k, h, w, c, d(wq), d(bq), d(wo), d(bo), fm0, obs0)
    x = np.concatenate((base, cvis, lvis), -1)
    q_ref = x @ wq[0, 0] + bq
    o_ref = (nn_rgb - nn_base) @ wo[0, 0] + bo
    agg = o_ref * ow[:, :, None, None, None] if weights else o_ref
    np.testing.assert_allclose(fm0.cpu().numpy()[..., :c], q_ref, atol=1e-5)
    np.testing.assert_allclose(obs0.cpu().numpy(), o_ref, atol=1e-5)
    np.testing.assert_allclose(fm0.cpu().numpy()[..., c:], agg.mean(1), atol=1e-5)


@pytest.mark.parametrize('k,weights', [(1, False), (4, False), (3, True)])
def test_obs_mean_into_slice(k, weights):
    rng = np.random.default_rng(10 + k)
    n, hw, c = 2, 33, 32
    obs = rng.standard_normal((n, k, hw, c)).astype(np.float32)
    ow = rng.random((n, k), dtype=np.float32) if weights else None
    fm = torch.full((n, hw, 2 * c), 5.0, device='cuda')
    C.obs_mean_forward(d(obs), d(ow), n, k, hw, c, fm.view(-1)[c:], 2 * c)
    ref = (obs * ow[:, :, None, None] if weights else obs).mean(1)
    got = fm.cpu().numpy()
    np.testing.assert_allclose(got[..., c:], ref, atol=1e-6)
    assert np.all(got[..., :c] == 5.0)


@pytest.mark.parametrize('with_base', [True, False])
def test_head(with_base):
    rng = np.random.default_rng(3)
    n, h, w, cd, cs = 2, 6, 5, 4, 32
    dec = rng.standard_normal((n, h, w, cd)).astype(np.float32)
    skip = rng.standard_normal((n, h, w, cs)).astype(np.float32)
    wk = rng.standard_normal((1, 1, cd + cs, 3)).astype(np.float32)
    b = rng.standard_normal(3).astype(np.float32)
    base = rng.random((n, h, w, 3), dtype=np.float32)
    pred = torch.empty(n, h, w, 3, device='cuda')
    C.head_forward(d(dec), cd, cd, d(skip), cs, cs, d(wk), d(b), d(base) if with_base else None, n, h, w, pred)
    ref = np.concatenate((dec, skip), -1) @ wk[0, 0] + b
    if with_base:
        ref = ref + base
    ref[:, 0, 0, :] = 0
    np.testing.assert_allclose(pred.cpu().numpy(), ref, atol=2e-5)


def _warp_case(n, uvh, uvw, hc, wc, seed, identity=False):
    rng = np.random.default_rng(seed)
    pred = rng.random((n, uvh, uvw, 3), dtype=np.float32); pred[:, 0, 0, :] = 0
    base = rng.random((n, uvh, uvw, 3), dtype=np.float32)
    if identity:
        jj, ii = np.meshgrid(np.arange(wc, dtype=np.float32), np.arange(hc, dtype=np.float32))
        warp = np.stack((jj / np.float32(wc), ii / np.float32(hc)), -1)[None].repeat(n, 0).astype(np.float32)
    else:
        warp = rng.random((n, hc, wc, 2), dtype=np.float32).astype(np.float16).astype(np.float32)
        warp[rng.random((n, hc, wc)) > 0.7] = 0
        warp[0, 0, 0] = (1.0, 0.5); warp[0, 0, 1] = (0.5, 1.0)          # x == W / y == H -> outside
        warp[0, 0, 2] = (np.float32(uvw - 1) / uvw, np.float32(uvh - 1) / uvh)   # last texel: cx, cy out of range
        warp[0, 0, 3] = (-0.001, 0.3)                                    # x in (-1, 0): fx = -1 is a zero tap
    return pred, base, warp


@pytest.mark.parametrize('identity', [False, True])
def test_warp_matches_oracle_and_indices_bit_exact(identity):
    n, uvh, uvw, hc, wc = 2, 32, 48, 16, 24
    if identity:
        hc, wc = uvh, uvw
    pred, base, warp = _warp_case(n, uvh, uvw, hc, wc, 0, identity)
    E = lambda: torch.empty(n, hc, wc, 3, device='cuda')
    pc, bc, fc = E(), E(), E()
    idx = torch.empty(n, hc, wc, 4, dtype=torch.int32, device='cuda')
    C.warp_forward(d(pred), d(base), d(warp), n, uvh, uvw, hc, wc, pc, bc, fc, idx)
    wpx = warp * np.float32([uvw, uvh])
    fx, fy, inside = T.resampler_indices(wpx, uvh, uvw)
    idx = idx.cpu().numpy()
    np.testing.assert_array_equal(idx[..., 0], fx)          # bit-exact integer UV indices
    np.testing.assert_array_equal(idx[..., 1], fy)
    np.testing.assert_array_equal(idx[..., 2], inside.astype(np.int32))
    base0 = base.copy(); base0[:, 0, 0, :] = 0
    fg = np.ones_like(pred); fg[:, 0, 0, :] = 0
    for got, src in ((pc, pred), (bc, base0), (fc, fg)):
        ref = T.resampler_naive(src, wpx)
        np.testing.assert_allclose(got.cpu().numpy(), ref, atol=1e-6, rtol=0)
    if identity:
        np.testing.assert_array_equal(pc.cpu().numpy(), pred)      # identity warp == no-op


@pytest.mark.parametrize('c', [4, 64])
def test_resampler_on_a_wide_map_matches_the_oracle(c):
    """nlt_resample_forward (the 64-channel stress point of SURVEY 8d): tfa.image.resampler's rule on a c-channel map, pixel
    units, no corner mask; same taps / order as the 3-channel warp, so within 1e-6 absolute of the oracle's float32 restatement (the bar of the 3-channel warp test above)."""
    n, h, w, hc, wc = 2, 24, 40, 16, 24
    _, _, warp = _warp_case(n, h, w, hc, wc, 5)
    rng = np.random.default_rng(c)
    data = rng.random((n, h, w, c), dtype=np.float32)
    wpx = (warp * np.float32([w, h])).astype(np.float32)
    got = C.resample_forward(d(data), d(wpx)).cpu().numpy()
    np.testing.assert_allclose(got, T.resampler_naive(data, wpx), atol=1e-6, rtol=0)
    with pytest.raises(C.NLTError):
        C.resample_forward(d(data[..., :3].copy()), d(wpx))


def test_warp_from_the_stores_equals_the_float_warp_bit_for_bit():
    """nlt_warp_forward_store: base from the uint8 diffuse store, the map from the fp16 uv2cam store (frame ids), against
    nlt_warp_forward on the float32 tensors `_load_data` would have produced -- identical bits and identical UV indices."""
    n, uvh, uvw, hc, wc, F = 3, 32, 48, 16, 24, 5
    rng = np.random.default_rng(11)
    pred, _, warp = _warp_case(n, uvh, uvw, hc, wc, 7)
    diffuse = torch.from_numpy(rng.integers(0, 256, (F, uvh, uvw, 3), dtype=np.uint8)).cuda()
    maps = torch.from_numpy(rng.random((F, hc, wc, 2), dtype=np.float32)).half()
    ids = torch.tensor([4, 0, 2], dtype=torch.int32)
    maps[ids.long()] = torch.from_numpy(warp).half()                  # (the test map is fp16-representable)
    maps = maps.cuda()
    base = (diffuse[ids.long().cuda()].double() / 255.0).float()
    E = lambda: torch.empty(n, hc, wc, 3, device='cuda')
    a, b = [E(), E(), E()], [E(), E(), E()]
    ia, ib = (torch.empty(n, hc, wc, 4, dtype=torch.int32, device='cuda') for _ in range(2))
    C.warp_forward(d(pred), base, maps[ids.long().cuda()].float(), n, uvh, uvw, hc, wc, *a, ia)
    C.warp_forward_store(d(pred), diffuse, maps, ids.cuda(), n, uvh, uvw, hc, wc, *b, ib)
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and torch.equal(ia, ib)


@pytest.mark.parametrize('oh,ow', [(8, 8), (32, 24), (5, 7), (16, 16)])
def test_resize(oh, ow):
    rng = np.random.default_rng(oh)
    x = rng.random((2, 16, 16, 3), dtype=np.float32)
    got = C.resize_bilinear_forward(d(x), oh, ow).cpu().numpy()
    np.testing.assert_allclose(got, T.resize_bilinear_naive(x, oh, ow), atol=1e-6)


def test_mul():
    rng = np.random.default_rng(0)
    a, b = rng.random(1003, dtype=np.float32), rng.random(1003, dtype=np.float32)
    np.testing.assert_array_equal(C.mul_forward(d(a), d(b)).cpu().numpy(), a * b)
