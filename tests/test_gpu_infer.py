"""-m gpu: the reference's INFERENCE mode -- nlt_test.infer -> Model.call(batch, 'test', obs_override=feat_agg)
(nlt/nlt_test.py:78-127, nlt/models/nlt.py:154-155,172-174) -- on the fused query-only plan (engine_infer.py,
csrc/front_ovr.hip, nlt_conv_forward_map) against the CPU oracle at BASELINE config 2's and config 3's sizes, against the
general layer-by-layer plan, replayed from the launch tape and through pipeline lanes.

Bars: rendered texels <= 1e-4 rel-L2 vs the oracle (measured ~1e-7), UV gather indices bit-exact, the two plans <= 1e-5 apart,
replays and lanes bit-identical to the eager pass."""
import os

import numpy as np
import pytest
import torch

from nlt_amd import _capi as C
from oracle import nlt_oracle as O
from oracle import tf_ops as T
from gpu_util import rel_l2, make_pair, to_device_batch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _agg_from_oracle(om, batches):
    """nlt_test.extract_feat on the oracle: mean over all frames of every level's observation features of x = rgb - base."""
    with torch.no_grad():
        feats = [om._call(torch.cat((b[1], b[2], b[3]), 3), [b[5] - b[1]], return_feats=True)[1] for b, _ in batches]
    return [torch.cat([f[l] for f in feats], 0).mean(0, keepdim=True) for l in range(len(feats[0]))]


@pytest.mark.parametrize('hw,n,add_base', [((64, 64), 2, True), ((72, 40), 1, True), ((128, 96), 3, False), ((36, 132), 2, True)])
def test_front_ovr_kernel_against_the_layerwise_arithmetic(hw, n, add_base):
    """nlt_front_ovr_forward alone (incl. sizes whose strips cross the right / bottom border) against float64 torch."""
    h, w = hw
    g = torch.Generator().manual_seed(h * 131 + w)
    R = lambda *s, lo=-0.5, hi=0.5: torch.rand(s, generator=g) * (hi - lo) + lo
    base, cvis, lvis = R(n, h, w, 3, lo=0, hi=1), R(n, h, w, 1, lo=0, hi=1), R(n, h, w, 1, lo=0, hi=1)
    wq0, bq0 = R(1, 1, 5, 16), R(16, lo=-0.1, hi=0.1)
    wqa, bqa = R(2, 2, 32, 16, lo=-0.2, hi=0.2), R(16, lo=-0.1, hi=0.1)
    wqb, bqb = R(2, 2, 16, 16, lo=-0.2, hi=0.2), R(16, lo=-0.1, hi=0.1)
    wq2, bq2 = R(2, 2, 32, 32, lo=-0.2, hi=0.2), R(32, lo=-0.1, hi=0.1)
    wh, bh = R(1, 1, 36, 3), R(3, lo=-0.1, hi=0.1)
    p1, s0, p2 = R(1, h // 2, w // 2, 16), R(1, h, w, 4), R(1, h // 4, w // 4, 32)
    z = lambda *s: torch.zeros(s)
    d = lambda t: t.cuda().contiguous()
    blob = C.front_pack_weights(d(wq0), d(bq0), d(z(1, 1, 3, 16)), d(z(16)), d(wqa), d(bqa), d(wqb), d(bqb), d(z(2, 2, 16, 16)), d(z(16)),
                                d(z(2, 2, 16, 16)), d(z(16)), d(wh), d(bh))
    blob2 = C.front_pack_l2_weights(d(wq2), d(bq2), d(z(2, 2, 16, 32)), d(z(32)))
    fm1 = torch.full((n, h // 2, w // 2, 32), -7.0, device='cuda')
    skip3 = torch.empty((n, h, w, 3), device='cuda')
    qtmp2 = torch.empty((n, h // 4, w // 4, 32), device='cuda')
    alpha = 0.3
    C.front_ovr_forward(d(base), d(cvis), d(lvis), n, h, w, blob, blob2, d(p1), d(s0), d(p2), add_base, alpha, fm1, 32, skip3, qtmp2)
    torch.cuda.synchronize()
    D = lambda t: t.double()
    lr = lambda t: torch.where(t > 0, t, alpha * t)
    q0 = torch.cat((base, cvis, lvis), -1).double() @ D(wq0)[0, 0]
    y1 = lr(T.conv2d_same(q0, D(wqa)[:, :, :16, :].contiguous(), z(16).double(), 2) + D(p1))
    q1 = lr(T.conv2d_same(y1, D(wqb), D(bqb), 1))
    t2 = lr(T.conv2d_same(q1, D(wq2)[:, :, :16, :].contiguous(), z(32).double(), 2) + D(p2))
    sk = q0 @ D(wh)[0, 0, 4:20, :] + D(s0)[..., :3] + (D(base) if add_base else 0)
    assert rel_l2(fm1[..., :16].cpu(), q1) <= 2e-6
    assert torch.all(fm1[..., 16:] == -7.0)                  # the given half of the interleaved map is not touched
    assert rel_l2(qtmp2.cpu(), t2) <= 2e-6
    assert rel_l2(skip3.cpu(), sk) <= 2e-6


@pytest.mark.parametrize('mode,c0,c1,cout,hw', [(C.CONV_K2S2, 32, 0, 64, (32, 48)), (C.CONV_K2S2, 256, 0, 256, (4, 4)),
                                              (C.DECONV_K2S2, 256, 0, 128, (2, 2)), (C.DECONV_K2S2, 128, 256, 64, (8, 8)),
                                              (C.DECONV_K2S2, 32, 64, 16, (24, 16)), (C.CONV1X1, 16, 0, 4, (16, 16))])
@pytest.mark.parametrize('frames', [1, 3])
def test_conv_forward_map_equals_conv_plus_map(mode, c0, c1, cout, hw, frames):
    """nlt_conv_forward_map: act(conv + bias + map) for a shared and a per-frame map, every wave tile, split-K."""
    from nlt_amd.networks.elements import Conv2D
    h, w = hw
    n = 3
    g = torch.Generator().manual_seed(c0 + 7 * cout)
    layer = Conv2D(cout, 1 if mode == C.CONV1X1 else 2, 2 if mode in (C.CONV_K2S2, C.DECONV_K2S2) else 1,
                   transpose=mode in (C.DECONV_K2S2, C.DECONV_K2S1))
    layer.build(c0 + c1, 'cuda', seed=5)
    layer.bias = (torch.rand(cout, generator=g) - 0.5).cuda()
    x0 = (torch.rand((n, h, w, c0), generator=g) - 0.5).cuda()
    x1 = (torch.rand((n, h, w, max(c1, 4)), generator=g) - 0.5).cuda()
    oh, ow = layer.out_hw(h, w)
    bmap = (torch.rand((frames if frames == 1 else n, oh, ow, cout), generator=g) - 0.5).cuda()
    ref = torch.empty((n, oh, ow, cout), device='cuda')
    C.conv_forward(mode, x0, c0, c0, x1 if c1 else None, c1, x1.shape[3] if c1 else 0, n, h, w, layer.kernel, layer.packed(c0, c1),
                   layer.bias, cout, ref, cout, act=False)
    want = torch.where(ref + bmap > 0, ref + bmap, 0.3 * (ref + bmap))
    ntiles = ((4 if mode == C.DECONV_K2S2 else 1) * cout + 15) // 16
    cases = [(1, 0), (1, 17), (1, 33), (4, 17), (16, 17)] + ([(1, 18), (8, 34)] if ntiles % 2 == 0 else [])
    for ks, hint in cases:
        out = torch.full((n, oh, ow, cout), float('nan'), device='cuda')
        C.conv_forward_map(mode, ks, x0, c0, c0, x1 if c1 else None, c1, x1.shape[3] if c1 else 0, n, h, w, layer.packed(c0, c1),
                           layer.bias, cout, out, cout, bmap, act=True, alpha=0.3, tile_hint=hint)
        torch.cuda.synchronize()
        assert rel_l2(out.cpu(), want.cpu()) <= 1e-6, (ks, hint)


@pytest.mark.parametrize('c,hw,n', [(8, (16, 32), 2), (8, (20, 24), 1), (16, (8, 16), 2), (16, (12, 40), 3)])
def test_dec_block_forward_map_equals_the_interleaved_block(c, hw, n):
    """nlt_dec_block_forward_map([x | query half], map) == nlt_dec_block_forward([x | query | given]) when the map is what the given
    half contributes: conv over the given rows of the first kernel + its bias (both on the GPU; <= 2e-6: a re-association)."""
    h, w = hw
    g = torch.Generator().manual_seed(c * 100 + h)
    R = lambda *s, sc=0.5: ((torch.rand(s, generator=g) - 0.5) * 2 * sc).cuda()
    x, fm = R(n, h, w, 2 * c), R(n, h, w, 8 * c)
    given = R(1, h, w, 4 * c)
    fm[..., 4 * c:] = given                                 # the interleaved map's given half: the same for every frame
    w2, b2, w1, b1 = R(2, 2, c, 10 * c, sc=0.2), R(c, sc=0.1), R(2, 2, c, c, sc=0.3), R(c, sc=0.1)
    ref = torch.empty((n, 2 * h, 2 * w, c), device='cuda')
    C.dec_block_forward(x, 2 * c, fm, 8 * c, n, h, w, w2, b2, w1, b1, c, 0.3, ref)
    wq = w2[..., :6 * c].contiguous()
    bmap = T.conv2d_transpose_same(given.cpu().double(), w2[..., 6 * c:].cpu().double(), b2.cpu().double(), 2).float().cuda().contiguous()
    out = torch.full_like(ref, float('nan'))
    C.dec_block_forward_map(x, fm, 8 * c, n, h, w, wq, w1, b1, c, 0.3, bmap, out)
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), ref.cpu()) <= 2e-6


@pytest.mark.parametrize('hw,n', [((16, 32), 2), ((20, 24), 1), ((64, 48), 3)])
def test_back_forward_map_equals_the_interleaved_kernel(hw, n):
    h2, w2 = hw
    g = torch.Generator().manual_seed(h2 * 7 + w2)
    R = lambda *s, sc=0.5: ((torch.rand(s, generator=g) - 0.5) * 2 * sc).cuda()
    x, fm1, skip3 = R(n, h2, w2, 8), R(n, h2, w2, 32), R(n, 2 * h2, 2 * w2, 3)
    given = R(1, h2, w2, 16)
    fm1[..., 16:] = given
    ws2, bs2, ws1, bs1, wh = R(2, 2, 4, 40, sc=0.2), R(4, sc=0.1), R(2, 2, 4, 4, sc=0.3), R(4, sc=0.1), R(1, 1, 36, 3, sc=0.3)
    ref = torch.empty((n, 2 * h2, 2 * w2, 3), device='cuda')
    C.back_forward(x, fm1, skip3, n, h2, w2, ws2, bs2, ws1, bs1, wh, 0.3, ref)
    wq = ws2[..., :24].contiguous()
    bmap = T.conv2d_transpose_same(given.cpu().double(), ws2[..., 24:].cpu().double(), bs2.cpu().double(), 2).float().cuda().contiguous()
    out = torch.full_like(ref, float('nan'))
    C.back_forward_map(x, fm1, 32, skip3, n, h2, w2, wq, ws1, bs1, wh, 0.3, bmap, out)
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), ref.cpu()) <= 2e-6


def _infer_vs_oracle(depth, uv, cam, n, identity_warp, seed, train_frames=2):
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    om, pm = make_pair(depth=depth, uv=uv, im=cam, seed=seed)
    pm.build('cuda')                                        # flat parameter bucket + pack registry: launch tapes need them
    train = [O.synth_batch(train_frames, uv, uv, cam, cam, cam, cam, k=1, seed=seed + 50)]
    agg = _agg_from_oracle(om, train)
    batch, nn = O.synth_batch(n, uv, uv, cam, cam, cam, cam, k=1, seed=seed + 100, identity_warp=identity_warp)
    with torch.no_grad():
        o_pred_c, _, _, o_vis = om.call(batch, 'test', obs_override=[f.expand(n, -1, -1, -1) for f in agg], nn_list=nn)
    db = to_device_batch(batch, nn)
    dagg = [f.cuda() for f in agg]
    outs = []
    for _ in range(3):                                      # plan-time trials, launch tape record, a replayed step
        p_pred_c, _, _, p_vis = pm.call(db, 'test', obs_override=dagg, want_indices=True)
        outs.append(p_vis['pred'].clone())
    torch.cuda.synchronize()
    assert pm.plan._ovr is not None and pm.plan.tape_replays >= 1
    assert torch.equal(outs[1], outs[2])                    # the replay reproduces the recorded pass bit for bit
    e_uv, e_cam = rel_l2(p_vis['pred'].cpu(), o_vis['pred']), rel_l2(p_pred_c.cpu(), o_pred_c)
    assert e_uv <= TOL and e_cam <= TOL, (e_uv, e_cam)
    fx, fy, inside = T.resampler_indices(o_vis['warp_px'].numpy(), uv, uv)
    idx = p_vis['uv_indices'].cpu().numpy()
    np.testing.assert_array_equal(idx[..., 0], fx)
    np.testing.assert_array_equal(idx[..., 1], fy)
    np.testing.assert_array_equal(idx[..., 2], inside.astype(np.int32))
    # the general (layer-by-layer) plan on the same maps
    pm.plan.fuse_override = False
    g_vis = pm.call(db, 'test', obs_override=dagg)[3]
    pm.plan.fuse_override = True
    assert rel_l2(p_vis['pred'].cpu(), g_vis['pred'].cpu()) <= 1e-5
    return om, pm, db, dagg, (e_uv, e_cam)


def test_infer_mode_small_sizes_and_depth_1024():
    _infer_vs_oracle(256, 64, 32, 2, False, seed=31)
    _infer_vs_oracle(1024, 256, 256, 2, True, seed=32)
    _infer_vs_oracle(64, 32, 32, 3, False, seed=33)


def test_infer_mode_at_config2_512_4_frames():
    """BASELINE config 2's shape (depth 256, 512^2 UV, 4 frames, identity warp) rendered the way nlt_test.infer renders."""
    _infer_vs_oracle(256, 512, 512, 4, True, seed=34)


def test_infer_mode_at_config3_uv_1024_4_frames():
    """BASELINE config 3's UV size (1024^2, 512^2 camera warp through fp16, 4 frames)."""
    _infer_vs_oracle(256, 1024, 512, 4, False, seed=35, train_frames=1)


def test_nlt_test_infer_end_to_end_with_lanes_and_new_feat_agg():
    """extract_feat -> infer on the GPU at 256^2: one batch at a time == 2 pipeline lanes (bit for bit); a second feat_agg
    (new maps, same plan) and an optimizer-style weight change rebuild the override state."""
    from nlt_amd import nlt_test
    uv, cam = 256, 128
    om, pm = make_pair(depth=256, uv=uv, im=cam, seed=41)
    pm.build('cuda')
    train = [O.synth_batch(2, uv, uv, cam, cam, cam, cam, k=1, seed=42 + i) for i in range(2)]
    tests_ = [O.synth_batch(2, uv, uv, cam, cam, cam, cam, k=1, seed=52 + i) for i in range(4)]
    ref_agg = _agg_from_oracle(om, train)
    agg = nlt_test.extract_feat(pm, [to_device_batch(b, nn) for b, nn in train])
    for a, r in zip(agg, ref_agg):
        assert rel_l2(a.cpu(), r) <= TOL
    dbs = [to_device_batch(b, nn) for b, nn in tests_]
    one = nlt_test.infer(pm, dbs, agg)
    with torch.no_grad():
        ref = om.call(tests_[3][0], 'test', obs_override=[f.expand(2, -1, -1, -1) for f in ref_agg], nn_list=tests_[3][1])[3]['pred']
    assert rel_l2(one[3]['pred'].cpu(), ref) <= TOL
    two = nlt_test.infer(pm, dbs, agg, lanes=2)
    for a, b in zip(one, two):
        assert torch.equal(a['pred'], b['pred']) and torch.equal(a['pred_camspc'], b['pred_camspc'])
    # another feat_agg on the same plan
    serial = pm.plan._ovr['serial']
    agg2 = [a * 0.5 + 0.01 for a in agg]
    out2 = nlt_test.infer(pm, dbs[:1], agg2)
    assert pm.plan._ovr['serial'] != serial
    with torch.no_grad():
        ref2 = om.call(tests_[0][0], 'test', obs_override=[(f * 0.5 + 0.01).expand(2, -1, -1, -1) for f in ref_agg],
                       nn_list=tests_[0][1])[3]['pred']
    assert rel_l2(out2[0]['pred'].cpu(), ref2) <= TOL
    # the weights change (as after an optimizer step): maps and derived kernels are re-made
    serial = pm.plan._ovr['serial']
    with torch.no_grad():
        pm.flat_params.mul_(1.01)
    pm.mark_weights_updated()
    out3 = nlt_test.infer(pm, dbs[:1], agg2)
    assert pm.plan._ovr['serial'] != serial
    pm.plan.fuse_override = False
    gen3 = nlt_test.infer(pm, dbs[:1], agg2)
    assert rel_l2(out3[0]['pred'].cpu(), gen3[0]['pred'].cpu()) <= 1e-5


@pytest.mark.parametrize('hw,n', [((64, 64), 3), ((72, 40), 2), ((36, 136), 2), ((256, 256), 2)])
def test_front_ovr_u8_reads_the_capture_store_like_the_float_kernel_reads_the_assembled_batch(hw, n):
    """nlt_front_ovr_forward_u8 (frame ids of the uint8 stores, 1 / 255 in registers) against nlt_front_ovr_forward on the float
    buffers `_load_data` would assemble (nlt/datasets/nlt.py:131-136): <= 1e-6 rel-L2 (fl(W / 255) . u vs W . fl(u / 255))."""
    h, w = hw
    g = torch.Generator().manual_seed(h * 31 + w)
    R = lambda *s, lo=-0.5, hi=0.5: (torch.rand(s, generator=g) * (hi - lo) + lo).cuda()
    F = 5
    U = lambda *s: torch.randint(0, 256, s, generator=g, dtype=torch.uint8).cuda()
    diffuse, cvis, lvis = U(F, h, w, 3), U(F, h, w), U(F, h, w)
    ids = torch.tensor([4, 0, 3][:n], dtype=torch.int32, device='cuda')
    z = lambda *s: torch.zeros(s, device='cuda')
    blob = C.front_pack_weights(R(1, 1, 5, 16), R(16, lo=-0.1, hi=0.1), z(1, 1, 3, 16), z(16), R(2, 2, 32, 16, lo=-0.2, hi=0.2),
                                R(16, lo=-0.1, hi=0.1), R(2, 2, 16, 16, lo=-0.2, hi=0.2), R(16, lo=-0.1, hi=0.1), z(2, 2, 16, 16), z(16),
                                z(2, 2, 16, 16), z(16), R(1, 1, 36, 3), R(3, lo=-0.1, hi=0.1))
    blob2 = C.front_pack_l2_weights(R(2, 2, 32, 32, lo=-0.2, hi=0.2), R(32, lo=-0.1, hi=0.1), z(2, 2, 16, 32), z(32))
    p1, s0, p2 = R(1, h // 2, w // 2, 16), R(1, h, w, 4), R(1, h // 4, w // 4, 32)
    outs = []
    for u8 in (False, True):
        fm1 = torch.full((n, h // 2, w // 2, 32), -7.0, device='cuda')
        skip3, qtmp2 = torch.empty((n, h, w, 3), device='cuda'), torch.empty((n, h // 4, w // 4, 32), device='cuda')
        if u8:
            C.front_ovr_forward_u8(diffuse, cvis, lvis, ids, n, h, w, blob, blob2, p1, s0, p2, True, 0.3, fm1, 32, skip3, qtmp2)
        else:
            f = lambda st: (st[ids.long()].double() / 255.0).float().contiguous()
            C.front_ovr_forward(f(diffuse), f(cvis).unsqueeze(-1), f(lvis).unsqueeze(-1), n, h, w, blob, blob2, p1, s0, p2, True, 0.3,
                                fm1, 32, skip3, qtmp2)
        torch.cuda.synchronize()
        outs.append((fm1, skip3, qtmp2))
    for a, b in zip(*outs):
        assert rel_l2(b.cpu(), a.cpu()) <= 1e-6
    assert torch.all(outs[1][0][..., 16:] == -7.0)
    with pytest.raises(C.NLTError):                       # 8-byte row pieces: w must be a multiple of 8
        C.front_ovr_forward_u8(diffuse[:, :, :w - 4].contiguous(), cvis[:, :, :w - 4].contiguous(), lvis[:, :, :w - 4].contiguous(), ids, n, h,
                               w - 4, blob, blob2, p1, s0, p2, True, 0.3, outs[1][0], 32, outs[1][1], outs[1][2])


def test_infer_on_store_resident_batches_equals_infer_on_the_assembled_float_batches():
    """nlt_test.infer over Dataset.load_batch(resident=True): the override plan's front launch reads the uint8 store by frame id
    (no float batch assembled), base / uv2cam are gathered from the stores -- against the eager float batches: rendered texels
    <= 1e-6 rel-L2, base_camspc and the UV gather indices bit-exact; tape replays and 2 lanes bit-identical to the first pass."""
    import nlt_amd
    from nlt_amd import nlt_test
    from nlt_amd.datasets import get_dataset_class
    from nlt_amd.datasets.synth import synthetic_store
    from nlt_amd.models import get_model_class
    uv, cam = 256, 128
    store = synthetic_store(9, uv, cam, seed=5, k=1)
    cfg = nlt_amd.make_config(depth=256, uvh=uv, uvw=uv, imh=cam, imw=cam, bs=2)
    pm = get_model_class('nlt')(cfg).build('cuda')
    pm.register_trainable()
    ds = tr = get_dataset_class('nlt')(cfg, 'train', store, k=1, ring=0)
    agg = nlt_test.extract_feat(pm, [tr.load_batch(store['ids'][i:i + 2]) for i in (0, 2)])
    id_lists = [store['ids'][i:i + 2] for i in (4, 6, 7)]
    eager = [ds.load_batch(i) for i in id_lists]
    res = [ds.load_batch(i, resident=True) for i in id_lists]
    assert res[0][2] is None
    a = [pm.call(b, 'test', obs_override=agg, want_indices=True) for b in eager]
    r0 = pm.plan.tape_replays
    b = [pm.call(b, 'test', obs_override=agg, want_indices=True) for b in res]
    torch.cuda.synchronize()
    assert pm.plan._ovr is not None
    for x, y in zip(a, b):
        assert rel_l2(y[3]['pred'].cpu(), x[3]['pred'].cpu()) <= 1e-6 and rel_l2(y[0].cpu(), x[0].cpu()) <= 1e-6
        assert torch.equal(x[3]['base_camspc'], y[3]['base_camspc']) and torch.equal(x[3]['uv_indices'], y[3]['uv_indices'])
    first = [y[3]['pred'].clone() for y in b]
    for _ in range(3):
        again = [pm.call(x, 'test', obs_override=agg) for x in res]
    torch.cuda.synchronize()
    assert pm.plan.tape_replays > r0
    for x, y in zip(first, again):
        assert torch.equal(x, y[3]['pred'])
    two = nlt_test.infer(pm, res, agg, lanes=2)
    for x, y in zip(first, two):
        assert torch.equal(x, y['pred'])
    # what the plan cannot take in place is materialised (general plan), same answer to 1e-5
    pm.plan.fuse_override = False
    gen = pm.call(res[0], 'test', obs_override=agg)
    assert rel_l2(gen[3]['pred'].cpu(), first[0].cpu()) <= 1e-5


def test_a_tiled_override_takes_the_fused_plan_and_a_per_frame_one_does_not():
    """The reference's call site hands Model.call `tf.tile(feat, (bs, 1, 1, 1))` (nlt/nlt_test.py:83-86): a materialised [N,h,w,C]
    tensor whose frames are copies.  The plan checks that on the device (once per tensor set) and takes the fused query-only plan --
    same texels, bit for bit, as with the [1,h,w,C] maps; frames that differ keep the general plan (and its answer)."""
    uv, cam, n = 128, 64, 3
    om, pm = make_pair(depth=256, uv=uv, im=cam, seed=61)
    pm.build('cuda')
    tr = [O.synth_batch(2, uv, uv, cam, cam, cam, cam, k=1, seed=62)]
    batch, nn = O.synth_batch(n, uv, uv, cam, cam, cam, cam, k=1, seed=63)
    agg = [a.cuda() for a in _agg_from_oracle(om, tr)]
    db = to_device_batch(batch, nn)
    one = pm.call(db, 'test', obs_override=agg)[3]['pred'].clone()
    serial = pm.plan._ovr['serial']
    tiled = [a.repeat(n, 1, 1, 1) for a in agg]
    got = pm.call(db, 'test', obs_override=tiled)[3]['pred'].clone()
    assert pm.plan._ovr['serial'] != serial                   # (other tensors: the state was rebuilt -- on the fused plan)
    assert torch.equal(got, one)
    r0 = pm.plan.tape_replays
    for _ in range(3):
        again = pm.call(db, 'test', obs_override=tiled)[3]['pred']
    assert pm.plan.tape_replays > r0 and torch.equal(again, one)
    # frames that differ: the general plan, checked against the oracle
    serial = pm.plan._ovr['serial']
    scale = torch.linspace(1.0, 1.3, n, device='cuda').view(n, 1, 1, 1)
    per_frame = [t * scale for t in tiled]
    out = pm.call(db, 'test', obs_override=per_frame)[3]['pred']
    assert pm.plan._ovr['serial'] == serial
    with torch.no_grad():
        ref = om.call(batch, 'test', obs_override=[t.cpu() for t in per_frame], nn_list=nn)[3]['pred']
    assert rel_l2(out.cpu(), ref) <= TOL
    # an in-place edit of one frame of the tiled tensors is seen (tensor version): back to the general plan
    tiled[0][1].mul_(1.5)
    out2 = pm.call(db, 'test', obs_override=tiled)[3]['pred']
    with torch.no_grad():
        ref2 = om.call(batch, 'test', obs_override=[t.cpu() for t in tiled], nn_list=nn)[3]['pred']
    assert rel_l2(out2.cpu(), ref2) <= TOL and rel_l2(out2.cpu(), one.cpu()) > 1e-4
