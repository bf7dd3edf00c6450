"""Pins oracle/buffers.py to tests/golden/buffer_assembly.npz -- outputs of the reference's OWN
Python (get_neighbors, grid_query_unstruct, normalize_uint, denormalize_float) run in the build
container by tests/golden/make_buffer_golden.py -- and checks the restated rows (cosines, diffuse
base, cv2.remap) through closed forms and size-independent properties."""
import os

import numpy as np
import pytest
import scipy.ndimage

from oracle import buffers as B

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'buffer_assembly.npz'))


def test_knn_matches_reference_get_neighbors():
    got = B.knn_indices(G['knn_ref'], G['knn_cand'], k=1)[:, 0]
    assert np.array_equal(got, G['knn_nn'])


def test_knn_exact_ties_first_minimum_wins():
    got = B.knn_indices(G['knn_lat'], G['knn_lat'], k=1)[:, 0]
    assert np.array_equal(got, G['knn_lat_nn'])


def test_knn_k_generalisation_and_shortfall():
    rng = np.random.default_rng(0)
    ref, cand = rng.normal(size=(6, 3)), rng.normal(size=(5, 3))
    k3 = B.knn_indices(ref, cand, k=3)
    assert np.array_equal(k3[:, 0], B.knn_indices(ref, cand, k=1)[:, 0])
    for p in range(6):
        d = np.linalg.norm(ref[p] - cand[k3[p]], axis=1)
        assert np.all(np.diff(d) >= 0)
    short = B.knn_indices(cand[:2], cand[:2], k=3)          # only one non-zero-distance candidate each
    assert np.array_equal(short, [[1, -1, -1], [0, -1, -1]])


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_uv_index_map_matches_reference_grid_query(tag):
    h, w = G['gq_%s_res' % tag]
    got, idx = B.uv_index_map(G['gq_%s_uvs' % tag], G['gq_%s_vals' % tag], (int(h), int(w)), return_index=True)
    ref = G['gq_%s_out' % tag]
    assert np.array_equal(got, ref)
    assert (idx < 0).any() and (idx >= 0).any()             # both the trusted and the filled branch are live


def test_l1_distance_matches_scipy_taxicab():
    rng = np.random.default_rng(3)
    occ = (rng.random((19, 23)) < 0.03).astype(np.uint8)
    occ[4, 5] = 1
    assert np.array_equal(B.l1_distance_to_occupied(occ), scipy.ndimage.distance_transform_cdt(1 - occ, metric='taxicab'))


def test_occupancy_indices_truncate_toward_zero():
    uvs = np.array([[0.0, 1.0], [1.0, 0.0], [0.999, 0.001], [-0.01, 0.5], [0.5, 1.25]])
    ri, ci, ok = B.occupancy_indices(uvs, 11, 21)
    assert ri.tolist()[:3] == [0, 10, 9] and ci.tolist()[:3] == [0, 20, 19]
    assert ci[3] == 0 and ok[3]                              # int(-0.2) == 0: truncation, not floor
    assert ri[4] == -2 and not ok[4]


def test_normalize_denormalize_match_reference():
    assert np.array_equal(B.normalize_uint(G['norm_u8']), G['norm_u8_out'])
    assert np.array_equal(B.normalize_uint(G['norm_u16']), G['norm_u16_out'])
    assert np.array_equal(B.denormalize_float(G['denorm_f']), G['denorm_f_out'])
    with pytest.raises(TypeError):
        B.normalize_uint(np.zeros(3, np.int32))


def test_cosine_map_closed_forms():
    locs = np.zeros((2, 3, 3)); normals = np.zeros((2, 3, 3)); normals[..., 2] = 2.0   # un-normalised normals
    locs[0, 1] = [1, 0, 0]
    valid = np.ones((2, 3), bool); valid[1, 2] = False
    occl = np.zeros((2, 3), bool); occl[0, 0] = True
    cam = [0, 0, 5]
    c = B.cosine_map(cam, locs, normals, valid)
    assert c[0, 0] == 1.0 and c[1, 2] == 0.0
    assert abs(c[0, 1] - 5 / np.sqrt(26)) < 1e-15
    l = B.cosine_map(cam, locs, normals, valid, occl)
    assert l[0, 0] == 0.0 and l[1, 1] == 1.0
    q = B.quantize_unit(np.array([[-0.5, 0.0, 0.5, 0.999, 1.0, 1.7]]))
    assert q.tolist() == [[0, 0, 127, 254, 255, 255]]        # truncation: 0.5*255 = 127.5 -> 127


def test_diffuse_base_formula():
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (5, 6, 7, 3), dtype=np.uint8)
    alb = B.albedo_from_frames(frames)
    assert alb.max() == 1.0 and alb.min() >= 0
    assert np.allclose(alb, frames.astype(np.float64).sum(0) / frames.astype(np.float64).sum(0).max(), rtol=1e-14)
    lvis = rng.integers(0, 256, (6, 7), dtype=np.uint8)
    d = B.diffuse_base(alb, lvis)
    assert d.dtype == np.uint8 and np.array_equal(d, np.floor(alb * (lvis / 255.0)[..., None] * 255).astype(np.uint8))


def test_remap_u8_integer_coordinates_are_an_index_gather():
    rng = np.random.default_rng(7)
    src = rng.integers(1, 256, (9, 12, 3), dtype=np.uint8)
    jj, ii = np.meshgrid(np.arange(12), np.arange(9))
    mapping = np.stack((jj / 12.0, ii / 9.0), -1)            # x = j exactly after * w
    out = B.remap_u8(src, mapping)
    exp = src.copy(); exp[0, 0] = 0                          # util.py:53-55 background sink
    assert np.array_equal(out, exp)
    assert np.array_equal(B.remap_u8(src, mapping, force_kbg=False), src)


def test_remap_u8_half_texel_and_border():
    src = np.zeros((4, 4), np.uint8); src[1, 1] = 200; src[1, 2] = 100
    m = np.zeros((1, 3, 2))
    m[0, 0] = [1.5 / 4, 1.0 / 4]                             # midway between (1,1) and (1,2): (200+100)/2
    m[0, 1] = [3.5 / 4, 3.0 / 4]                             # right tap is outside -> constant 0 border
    m[0, 2] = [-0.5 / 4, 1.0 / 4]
    src[3, 3] = 80
    out = B.remap_u8(src, m, force_kbg=False)
    assert out.tolist() == [[150, 40, 0]]


def test_remap_f32_close_to_u8_path():
    rng = np.random.default_rng(9)
    src = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    mapping = rng.random((10, 10, 2))
    a = B.remap_u8(src, mapping).astype(np.float64)
    b = B.remap_f32(src.astype(np.float32), mapping).astype(np.float64)
    assert np.abs(a - b).max() <= 0.5 + 1e-3                 # same 1/32-pixel coordinates; only the final rounding differs


def test_assemble_batch_semantics():
    rng = np.random.default_rng(11)
    F, H, W = 5, 4, 6
    store = {'diffuse': rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8), 'rgb': rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8),
             'cvis': rng.integers(0, 256, (F, H, W), dtype=np.uint8), 'lvis': rng.integers(0, 256, (F, H, W), dtype=np.uint8)}
    b = B.assemble_batch(store, [3, 0], [[1, -1], [4, 2]])
    assert b['base'].dtype == np.float32 and b['cvis'].shape == (2, H, W, 1) and b['nn_rgb'].shape == (2, 2, H, W, 3)
    assert np.array_equal(b['base'][0], (store['diffuse'][3] / 255.0).astype(np.float32))
    assert not b['nn_base'][0, 1].any() and np.array_equal(b['nn_rgb'][1, 0], (store['rgb'][4] / 255.0).astype(np.float32))
    assert not B.assemble_batch(store, [3], [[1]], mode='test')['rgb'].any()


def test_channel_schedule_matches_reference_gen_feat_n():
    """nlt/util/net.py:18-56 run by the fixture script: oracle AND product schedule, 50+ (min, max, final) triples."""
    from oracle import nlt_oracle as O
    from nlt_amd.util.net import gen_feat_n
    off = 0
    for (a, b, c), n in zip(G['feat_n_args'].tolist(), G['feat_n_len'].tolist()):
        ref = G['feat_n_flat'][off:off + n].tolist(); off += n
        assert O.gen_feat_n(a, b, c) == ref, (a, b, c)
        assert gen_feat_n(a, b, c) == ref, (a, b, c)
