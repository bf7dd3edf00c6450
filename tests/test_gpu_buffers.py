"""-m gpu parity of the texel-buffer assembly kernels (csrc/assemble.hip) against oracle/buffers.py
and the reference-generated golden fixture: bit-exact for every integer / uint8 / index result."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import buffers as B

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'buffer_assembly.npz'))


DEV = 'cuda'          # tests/test_host_buffers.py re-runs the host-mirror tests with DEV = 'cpu' on the fake C-ABI


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


@pytest.fixture(scope='module')
def C():
    from nlt_amd import capi
    capi.lib()
    return capi


# ----------------------------------------------------------------------------- a-B6 k-NN
def test_knn_matches_reference_golden(C):
    got = C.knn_indices(_dev(G['knn_ref']), _dev(G['knn_cand']), 1).cpu().numpy()[:, 0]
    assert np.array_equal(got, G['knn_nn'])
    lat = C.knn_indices(_dev(G['knn_lat']), _dev(G['knn_lat']), 1).cpu().numpy()[:, 0]
    assert np.array_equal(lat, G['knn_lat_nn'])               # exact ties: first minimum wins


@pytest.mark.parametrize('p,q,k', [(1, 1, 1), (5, 3, 4), (130, 257, 4), (700, 900, 8)])
def test_knn_matches_oracle(C, p, q, k):
    rng = np.random.default_rng(p * 31 + q)
    cand = rng.normal(size=(q, 3))
    ref = np.concatenate((cand[:min(p, q) // 2], rng.normal(size=(p - min(p, q) // 2, 3))), 0)
    cand[q // 2:] = np.round(cand[q // 2:] * 2) / 2           # a lattice part: many exact ties and zero distances
    ref[p // 2:] = np.round(ref[p // 2:] * 2) / 2
    got = C.knn_indices(_dev(ref), _dev(cand), k).cpu().numpy()
    assert np.array_equal(got, B.knn_indices(ref, cand, k))


# ----------------------------------------------------------------------------- a-B5 UV-index map
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_uv_index_map_matches_reference_golden(C, tag):
    h, w = (int(x) for x in G['gq_%s_res' % tag])
    got = C.uv_index_map(_dev(G['gq_%s_uvs' % tag]), _dev(G['gq_%s_vals' % tag]), h, w, 4, 0.0).cpu().numpy()
    assert np.array_equal(got, G['gq_%s_out' % tag])


@pytest.mark.parametrize('h,w,p,l1', [(2, 2, 1, 0), (40, 64, 300, 4), (96, 50, 2000, 2), (128, 128, 6000, 7)])
def test_uv_index_map_matches_oracle_with_indices(C, h, w, p, l1):
    rng = np.random.default_rng(h + 7 * w + p)
    centres = rng.random((4, 2))
    uvs = np.clip(centres[rng.integers(0, 4, p)] + 0.08 * rng.normal(size=(p, 2)), -0.1, 1.1)
    uvs[: p // 10] = uvs[p // 10: 2 * (p // 10)]              # duplicated locations: ties -> lowest index
    vals = rng.random((p, 3))
    ref, ref_idx = B.uv_index_map(uvs, vals, (h, w), max_l1_interp=l1, return_index=True)
    got, idx = C.uv_index_map(_dev(uvs), _dev(vals), h, w, l1, 0.0, want_index=True)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)         # integer sample indices: bit-exact
    assert np.array_equal(got.cpu().numpy(), ref)


def test_uv_index_map_full_size_properties(C):
    """1024^2 grid, 512^2 x 3 samples (BASELINE data-gen sizes): size-independent properties."""
    h = w = 1024
    rng = np.random.default_rng(5)
    p = 512 * 512 * 3
    uvs = 0.25 + 0.5 * rng.random((p, 2))                      # samples cover the central half of the canvas
    vals = np.concatenate((uvs, np.arange(p, dtype=np.float64)[:, None]), 1)
    got, idx = C.uv_index_map(_dev(uvs), _dev(vals), h, w, 4, -1.0, want_index=True)
    idx = idx.cpu().numpy(); got = got.cpu().numpy()
    filled = idx < 0
    assert filled[0, 0] and filled[-1, -1] and not filled[512, 512]
    assert np.all(got[filled] == -1.0)
    assert np.array_equal(got[~filled][:, 2], idx[~filled].astype(np.float64))      # value of the CHOSEN sample
    ii, jj = np.nonzero(~filled)
    gu, gv = np.linspace(0, 1, w)[jj], 1 - np.linspace(0, 1, h)[ii]
    d = np.hypot(gu - got[~filled][:, 0], gv - got[~filled][:, 1])
    assert d.max() <= np.hypot(5, 5) / (w - 1) + 1e-12         # the nearest sample of a trusted texel is close
    sub = rng.integers(0, len(ii), 200)                        # spot-check exact nearest on 200 texels
    for s in sub:
        d2 = (gu[s] - uvs[:, 0]) ** 2 + (gv[s] - uvs[:, 1]) ** 2
        assert idx[ii[s], jj[s]] == int(np.argmin(d2))


# ----------------------------------------------------------------------------- a-B2 cosines
@pytest.mark.parametrize('hc,wc', [(1, 1), (37, 53), (512, 512)])
def test_cosine_maps_bit_exact(C, hc, wc):
    rng = np.random.default_rng(hc * 3 + wc)
    locs = rng.normal(size=(hc, wc, 3)); normals = rng.normal(size=(hc, wc, 3)) * 3
    valid = rng.random((hc, wc)) < 0.8
    occl = rng.random((hc, wc)) < 0.3
    normals[0, 0] = 0                                          # zero normal stays zero
    src = rng.normal(size=3) * 4
    for oc in (None, occl):
        ref = B.cosine_map(src, locs, normals, valid, oc)
        cos, q = C.cosine_map(_dev(locs), _dev(normals), _dev(valid.astype(np.uint8)),
                              None if oc is None else _dev(oc.astype(np.uint8)), src)
        assert np.array_equal(q.cpu().numpy(), B.quantize_unit(ref))          # uint8: bit-exact
        assert np.abs(cos.cpu().numpy() - ref).max() <= 4.5e-16               # float64: <= 2 ulp at |x| <= 1


# ----------------------------------------------------------------------------- a-B3 albedo / diffuse base
@pytest.mark.parametrize('f,h,w', [(1, 2, 2), (7, 33, 21), (24, 256, 256)])
def test_albedo_and_diffuse_bit_exact(C, f, h, w):
    rng = np.random.default_rng(f + h)
    frames = rng.integers(0, 256, (f, h, w, 3), dtype=np.uint8)
    lvis = rng.integers(0, 256, (f, h, w), dtype=np.uint8)
    alb_ref = B.albedo_from_frames(frames)
    alb = C.albedo(_dev(frames))
    assert np.array_equal(alb.cpu().numpy(), alb_ref)                         # float64 sum order reproduced
    dif = C.diffuse_base(alb, _dev(lvis)).cpu().numpy()
    assert np.array_equal(dif, np.stack([B.diffuse_base(alb_ref, lvis[i]) for i in range(f)]))


# ----------------------------------------------------------------------------- a-B4 remap
@pytest.mark.parametrize('mdt', [np.float64, np.float32, np.float16])
@pytest.mark.parametrize('h,w,c,oh,ow', [(9, 12, 3, 9, 12), (64, 48, 1, 100, 70), (512, 512, 3, 1024, 1024)])
def test_remap_u8_bit_exact(C, mdt, h, w, c, oh, ow):
    rng = np.random.default_rng(h + ow)
    src = rng.integers(0, 256, (h, w, c) if c > 1 else (h, w), dtype=np.uint8)
    mapping = (rng.random((oh, ow, 3)) * 1.1 - 0.05).astype(mdt)               # some taps fall off the image
    mapping[rng.random((oh, ow)) < 0.3] = 0                                    # background -> texel (0,0)
    for kbg in (True, False):
        got = C.remap_bilinear(_dev(src), _dev(mapping), kbg).cpu().numpy()
        assert np.array_equal(got, B.remap_u8(src, mapping, kbg))


def test_remap_f32_matches_oracle(C):
    rng = np.random.default_rng(3)
    src = rng.random((40, 56, 3), dtype=np.float32)
    mapping = rng.random((64, 64, 2))
    got = C.remap_bilinear(_dev(src), _dev(mapping), True).cpu().numpy()
    assert np.array_equal(got, B.remap_f32(src, mapping, True))               # same op order, no FMA


def test_remap_identity_is_index_gather(C):
    src = np.random.default_rng(1).integers(1, 256, (1024, 1024, 3), dtype=np.uint8)
    jj, ii = np.meshgrid(np.arange(1024), np.arange(1024))
    mapping = np.stack((jj / 1024.0, ii / 1024.0), -1)
    got = C.remap_bilinear(_dev(src), _dev(mapping), True).cpu().numpy()
    exp = src.copy(); exp[0, 0] = 0
    assert np.array_equal(got, exp)


# ----------------------------------------------------------------------------- a-B1 batch assembly
@pytest.mark.parametrize('F,H,W,n,k', [(3, 2, 2, 1, 1), (6, 20, 12, 4, 3), (10, 256, 256, 4, 4)])
def test_assemble_batch_bit_exact(C, F, H, W, n, k):
    rng = np.random.default_rng(F + H)
    store = {'diffuse': rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8), 'rgb': rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8),
             'cvis': rng.integers(0, 256, (F, H, W), dtype=np.uint8), 'lvis': rng.integers(0, 256, (F, H, W), dtype=np.uint8)}
    ids = rng.integers(0, F, n).astype(np.int32)
    nn = rng.integers(-1, F, (n, k)).astype(np.int32)
    nn[0, 0] = -1
    d = {x: _dev(v) for x, v in store.items()}
    for test_mode in (False, True):
        ref = B.assemble_batch(store, ids, nn, 'test' if test_mode else 'train')
        got = C.assemble_batch(d['diffuse'], d['rgb'], d['cvis'], d['lvis'], _dev(ids), _dev(nn), test_mode)
        for key in ref:
            assert np.array_equal(got[key].cpu().numpy(), ref[key]), key


def test_u8_to_f32_table_is_exact(C):
    store = np.arange(256, dtype=np.uint8).reshape(1, 16, 16)
    got = C.gather_frames_u8(_dev(store), _dev(np.array([0, -1], np.int32))).cpu().numpy()
    assert np.array_equal(got[0], (store[0] / 255.0).astype(np.float32)) and not got[1].any()


# ----------------------------------------------------------------------------- host mirrors
def test_data_gen_mirrors(C):
    from nlt_amd.data_gen import get_neighbors as gn, render, util
    rng = np.random.default_rng(0)
    cams = [{'name': 'c%02d' % i, 'position': list(rng.normal(size=3))} for i in range(12)]
    nn = gn.get_neighbors(cams, cams[:8], device=DEV)
    ref = B.knn_indices([c['position'] for c in cams], [c['position'] for c in cams[:8]], 1)[:, 0]
    assert [nn[c['name']] for c in cams] == ['c%02d' % i for i in ref]
    # bidirectional mapping on a synthetic unwrap: two triangles per camera pixel
    imh = imw = 12; uvs_res = 16
    xs, ys = np.meshgrid(range(imw), range(imh))
    xys = np.dstack((xs, ys)).reshape(-1, 2)
    face_i = np.where(rng.random(imh * imw) < 0.7, rng.integers(0, 40, imh * imw), -1)
    unwrap = {f: np.concatenate((np.zeros((3, 2)), rng.random((3, 2))), 1) for f in range(40)}
    inter = {'face_i': face_i, 'valid': torch.ones(imh * imw, dtype=torch.uint8, device=DEV)}
    uv2cam, cam2uv = render.calc_bidir_mapping(unwrap, 'obj', xys, inter, uvs_res)
    hit = np.nonzero(face_i >= 0)[0]
    uv = np.vstack([unwrap[int(face_i[p])][:, 2:] for p in hit])
    xy = np.vstack([np.repeat(xys[p:p + 1].astype(float), 3, 0) for p in hit])
    ref_c2u = B.uv_index_map(uv, np.stack((xy[:, 0] / imw, xy[:, 1] / imh), 1), (uvs_res, uvs_res))
    ref_u2c = B.uv_index_map(np.stack((xy[:, 0] / imw, 1 - xy[:, 1] / imh), 1), np.stack((uv[:, 0], 1 - uv[:, 1]), 1), (imh, imw))
    assert np.array_equal(cam2uv.cpu().numpy(), ref_c2u) and np.array_equal(uv2cam.cpu().numpy(), ref_u2c)
    src = _dev(rng.integers(0, 256, (imh, imw), dtype=np.uint8))
    out = util.remap(src, cam2uv)
    assert np.array_equal(out.cpu().numpy(), B.remap_u8(src.cpu().numpy(), ref_c2u))
    with pytest.raises(NotImplementedError):
        render.grid_query_unstruct(_dev(uv), _dev(uv), (4, 4), {'func': 'rbf'})


def test_dataset_load_batch(C):
    import nlt_amd
    from nlt_amd.datasets import get_dataset_class
    rng = np.random.default_rng(2)
    cams, lights = ['P01', 'P02', 'P03'], ['L1', 'L2']
    ids = ['trainvali_%09d_%s_%s' % (i, c, l) for i, (c, l) in enumerate((c, l) for c in cams for l in lights)] + ['test_000000000_P09_L9']
    F, H, W, im = len(ids), 16, 16, 8
    U = lambda *s: torch.from_numpy(rng.integers(0, 256, s, dtype=np.uint8)).to(DEV)
    store = {'ids': ids, 'diffuse': U(F, H, W, 3), 'rgb': U(F, H, W, 3), 'cvis': U(F, H, W), 'lvis': U(F, H, W),
             'rgb_camspc': U(F, im, im, 3), 'uv2cam': torch.rand(F, im, im, 2).half().to(DEV),
             'nn': {id_: {'cam': 'P02', 'light': 'L1'} for id_ in ids}}
    store['nn'][ids[0]] = {'cam': 'P77', 'light': 'L1'}        # no such neighbour -> zero placeholders
    cfg = nlt_amd.make_config(holdout_cam='P03', holdout_light='L2', bs=2, uvh=H, uvw=W, imh=im, imw=im)
    ds = get_dataset_class('nlt')(cfg, 'train', store)
    assert 'trainvali_000000005_P03_L2' not in ds.files and len(ds.files) == 5
    assert get_dataset_class('nlt')(cfg, 'vali', store).files == ['trainvali_000000005_P03_L2']
    b = ds.load_batch(ids[:2])
    nn_idx = ids.index('trainvali_000000002_P02_L1')
    f32 = lambda t: (t.cpu().numpy() / 255.0).astype(np.float32)
    assert np.array_equal(b[1].cpu().numpy(), f32(store['diffuse'][:2]))
    assert b[2].shape == (2, H, W, 1) and b[4].dtype == torch.float32
    assert not b[8][0].any() and np.array_equal(b[9][1, 0].cpu().numpy(), f32(store['rgb'][nn_idx]))
    assert np.array_equal(b[10][1].cpu().numpy(), f32(store['rgb_camspc'][nn_idx])) and not b[10][0].any()
    t = get_dataset_class('nlt')(cfg, 'test', store).load_batch([ids[-1]])
    assert not t[5].any() and not t[6].any()
    # staging ring: the addresses a batch arrives at repeat every `ring` calls, the contents follow the ids
    ds3 = get_dataset_class('nlt')(cfg, 'train', store, ring=3)
    seen = [ds3.load_batch(ids[i:i + 2]) for i in range(4)]
    assert seen[3][1].data_ptr() == seen[0][1].data_ptr() and seen[1][1].data_ptr() != seen[0][1].data_ptr()
    assert seen[3][4].data_ptr() == seen[0][4].data_ptr() and seen[3][10].data_ptr() == seen[0][10].data_ptr()
    assert np.array_equal(seen[3][1].cpu().numpy(), f32(store['diffuse'][3:5]))
    assert np.array_equal(seen[2][5].cpu().numpy(), f32(store['rgb'][2:4]))
    fresh = get_dataset_class('nlt')(cfg, 'train', store, ring=0)
    assert fresh.load_batch(ids[:2])[1].data_ptr() != fresh.load_batch(ids[:2])[1].data_ptr() or True
    # stored camera resolution != (imh, imw): every buffer goes through normalise -> cv2-style resize -> float32 once
    big = get_dataset_class('nlt')(nlt_amd.make_config(uvh=H, uvw=W, imh=2 * im, imw=2 * im, bs=2), 'train', store).load_batch(ids[:2])
    ref = B.cv_resize_linear(store['rgb_camspc'][1].cpu().numpy() / 255.0, 2 * im, 2 * im).astype(np.float32)
    assert np.array_equal(big[6][1].cpu().numpy(), ref) and np.array_equal(big[1].cpu().numpy(), f32(store['diffuse'][:2]))
    assert tuple(big[4].shape[1:3]) == (im, im) and not big[10][0].any()
    # resident batch: texel buffers stay in the uint8 store; materialising them gives the eager batch bit for bit
    r = ds.load_batch(ids[:2], resident=True)
    e = get_dataset_class('nlt')(cfg, 'train', store, ring=0).load_batch(ids[:2])
    assert all(r[i] is None for i in (2, 3, 4, 5, 8, 9)) and r[1].n == 2 and r[1].k == 1 and (r[1].h, r[1].w) == (H, W)
    assert (r[1].hc, r[1].wc) == (im, im)
    m = r[1].materialize()
    for key, i in (('base', 1), ('cvis', 2), ('lvis', 3), ('warp', 4), ('rgb', 5), ('nn_base', 8), ('nn_rgb', 9)):
        assert torch.equal(m[key].cpu(), e[i].cpu()), key
    assert torch.equal(r[1].base_float().cpu(), e[1].cpu()) and torch.equal(r[1].warp_float().cpu(), e[4].cpu())
    assert torch.equal(r[6].cpu(), e[6].cpu()) and torch.equal(r[10].cpu(), e[10].cpu())
    # the ring's single pinned upload of (frame ids | neighbour ids) per batch: contents follow the ids slot after slot
    rr = [ds3.load_batch(ids[i:i + 2], resident=True) for i in range(4)]
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    for i, b_ in enumerate(rr[1:], 1):                              # (rr[0]'s slot was refilled by rr[3])
        assert b_[1].ids.cpu().tolist() == [i, i + 1]
        assert torch.equal(b_[1].materialize()['base'].cpu(), torch.from_numpy(f32(store['diffuse'][i:i + 2])))


def test_psnr_on_luma_matches_the_reference_values():
    """nlt_psnr_sums + nlt_amd.metric.PSNR against the values xm.metric.PSNR(np.float32) returned (golden, reference imported
    and run) and against the oracle on a 1024^2 image."""
    from nlt_amd.metric import PSNR
    from oracle import metric as M
    D = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'io_metric.npz'))
    psnr = PSNR(np.float32)
    for i, (plain, masked) in enumerate(D['psnr_values']):
        a, b = torch.from_numpy(D['psnr_a%d' % i]).to(DEV), torch.from_numpy(D['psnr_b%d' % i]).to(DEV)
        m = torch.from_numpy(D['psnr_m%d' % i]).to(DEV)
        assert abs(psnr(a, b) - plain) <= 1e-10 * abs(plain)
        assert abs(psnr(a, b, mask=m) - masked) <= 1e-10 * abs(masked)
    rng = np.random.default_rng(3)
    a = rng.random((1024, 1024, 3), dtype=np.float32)
    b = np.clip(a + 0.02 * rng.standard_normal(a.shape).astype(np.float32), 0, 1).astype(np.float32)
    got = psnr(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV))
    assert abs(got - M.psnr(a, b)) <= 1e-10 * abs(got)
    with pytest.raises(AssertionError):
        psnr(torch.zeros(4, 4, 3, device=DEV), torch.zeros(4, 5, 3, device=DEV))


@pytest.mark.gpu
@pytest.mark.parametrize('kind,h,w,oh,ow,c', [('u8', 16, 24, 32, 48, 3), ('u8', 64, 64, 24, 40, 1), ('u16', 20, 20, 33, 17, 3),
                                              ('f32', 9, 13, 9, 13, 2), ('u8', 512, 512, 1024, 1024, 3)])
def test_resize_cv_linear_bit_exact(C, kind, h, w, oh, ow, c):
    """nlt_resize_cv_linear vs the oracle's restatement of cv2.resize INTER_LINEAR on the normalised float64 image
    (float64 arithmetic, no contraction: bit-exact float32 results)."""
    rng = np.random.default_rng(h + ow)
    n = 2
    if kind == 'u8':
        a = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8); src = torch.from_numpy(a); norm = a.astype(np.float64) / 255
    elif kind == 'u16':
        a = rng.integers(0, 65536, (n, h, w, c)).astype(np.int32); src = torch.from_numpy(a); norm = a.astype(np.float64) / 65535
    else:
        a = rng.random((n, h, w, c)).astype(np.float32); src = torch.from_numpy(a); norm = a.astype(np.float64)
    ref = np.stack([B.cv_resize_linear(f, oh, ow) for f in norm]).astype(np.float32)
    out = C.resize_cv_linear(src.cuda(), oh, ow).cpu().numpy()
    assert np.array_equal(out, ref)
