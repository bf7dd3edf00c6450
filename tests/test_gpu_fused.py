"""-m gpu: the fused inference ends (csrc/fused.hip: front = L0 folded into L1 + observation means,
back = last expanding block + head) against the CPU oracle, stand-alone and inside Model.call, and
against the layer-by-layer plan at the bench size.  Tolerance: rel-L2 <= 1e-4 vs the oracle (BASELINE.json),
<= 1e-5 between the two plans (fp32 re-association only)."""
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from nlt_amd.engine import OpTimer
from oracle import nlt_oracle as O
from oracle import tf_ops as T
from gpu_util import rel_l2, make_pair, to_device_batch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _weights(om):
    W = om.numpy_weights()
    (wq0, bq0), = W['query'][0]; (wo0, bo0), = W['obs'][0]
    (wqa, bqa), (wqb, bqb) = W['query'][1]; (woa, boa), (wob, bob) = W['obs'][1]
    (wh, bh), = W['query'][-1]
    up = W['query'][len(om.layers) - 2]
    return (wq0, bq0, wo0, bo0, wqa, bqa, wqb, bqb, woa, boa, wob, bob, wh, bh), up


@pytest.mark.parametrize('n,h,w,k,add_base', [(1, 64, 64, 2, True), (2, 40, 56, 3, False), (1, 16, 32, 1, True),
                                              (1, 2, 2, 1, True), (2, 128, 256, 4, True), (1, 36, 44, 5, True)])
def test_front_and_back_kernels_vs_oracle_layers(n, h, w, k, add_base):
    om = O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32, seed=n + h + k)
    rng = np.random.default_rng(h * 7 + w)
    U = lambda *s: torch.from_numpy(rng.random(s, dtype=np.float32))
    base, cvis, lvis = U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1)
    nn = [(U(n, h, w, 3), U(n, h, w, 3)) for _ in range(k)]
    fw, ((w_s2, b_s2), (w_s1, b_s1)) = _weights(om)
    wh, bh = fw[12], fw[13]
    nl = len(om.layers)
    x = U(n, h // 2, w // 2, 8)
    with torch.no_grad():
        o0 = [O.apply_layer(om.layers[0], om.wo[0], r - b) for b, r in nn]
        fm0 = torch.cat((O.apply_layer(om.layers[0], om.wq[0], torch.cat((base, cvis, lvis), 3)), torch.stack(o0, -1).mean(-1)), -1)
        o1 = [O.apply_layer(om.layers[1], om.wo[1], o) for o in o0]
        fm1_ref = torch.cat((O.apply_layer(om.layers[1], om.wq[1], fm0), torch.stack(o1, -1).mean(-1)), -1)
        skip_ref = fm0 @ torch.from_numpy(wh[0, 0, 4:, :]) + torch.from_numpy(bh) + (base if add_base else 0)
        d = O.apply_layer(om.layers[nl - 2], om.wq[nl - 2], torch.cat((x, fm1_ref), -1))
        pred_ref = T.set_left_top_corner(d @ torch.from_numpy(wh[0, 0, :4, :]) + skip_ref, 0)
    dev = lambda a: torch.as_tensor(a).cuda().contiguous()
    blob = C.front_pack_weights(*[dev(a) for a in fw])
    fm1 = torch.full((n, h // 2, w // 2, 32), float('nan'), device='cuda')
    obs1 = torch.full((n, k, h // 2, w // 2, 16), float('nan'), device='cuda')
    skip3 = torch.full((n, h, w, 3), float('nan'), device='cuda')
    C.front_forward(dev(base), dev(cvis), dev(lvis), dev(torch.stack([r for _, r in nn], 1)), dev(torch.stack([b for b, _ in nn], 1)),
                    n, k, h, w, blob, add_base, 0.3, fm1, obs1, skip3)
    torch.cuda.synchronize()
    assert not torch.isnan(fm1).any() and not torch.isnan(obs1).any() and not torch.isnan(skip3).any()
    assert rel_l2(fm1.cpu(), fm1_ref) <= 1e-5
    assert rel_l2(obs1.cpu(), torch.stack(o1, 1)) <= 1e-5
    assert rel_l2(skip3.cpu(), skip_ref) <= 1e-5
    if k <= 4 and h % 4 == 0 and w % 4 == 0:                       # the variant that also runs level 2's stride-2 convs
        W = om.numpy_weights()
        (wqa2, bqa2), _ = W['query'][2]; (woa2, boa2), _ = W['obs'][2]
        with torch.no_grad():
            q2 = T.leaky_relu(T.conv2d_same(fm1_ref, torch.from_numpy(wqa2), torch.from_numpy(bqa2), 2))
            o2 = torch.stack([T.leaky_relu(T.conv2d_same(o, torch.from_numpy(woa2), torch.from_numpy(boa2), 2)) for o in o1], 1)
        blob2 = C.front_pack_l2_weights(dev(wqa2), dev(bqa2), dev(woa2), dev(boa2))
        fm1b = torch.full_like(fm1, float('nan')); skip3b = torch.full_like(skip3, float('nan'))
        qt = torch.full((n, h // 4, w // 4, 32), float('nan'), device='cuda')
        ot = torch.full((n, k, h // 4, w // 4, 32), float('nan'), device='cuda')
        C.front2_forward(dev(base), dev(cvis), dev(lvis), dev(torch.stack([r for _, r in nn], 1)), dev(torch.stack([b for b, _ in nn], 1)),
                         n, k, h, w, blob, blob2, add_base, 0.3, fm1b, skip3b, qt, ot)
        torch.cuda.synchronize()
        assert torch.equal(fm1b, fm1) and torch.equal(skip3b, skip3)                 # same arithmetic as the plain front kernel
        assert not torch.isnan(qt).any() and not torch.isnan(ot).any()
        assert rel_l2(qt.cpu(), q2) <= 1e-5 and rel_l2(ot.cpu(), o2) <= 1e-5
    pred = torch.full((n, h, w, 3), float('nan'), device='cuda')
    C.back_forward(dev(x), dev(fm1_ref), dev(skip_ref), n, h // 2, w // 2, dev(w_s2), dev(b_s2), dev(w_s1), dev(b_s1), dev(wh),
                   0.3, pred)
    torch.cuda.synchronize()
    assert not torch.isnan(pred).any()
    assert rel_l2(pred.cpu(), pred_ref) <= 1e-5
    assert not pred[:, 0, 0].any()


def _labels(pm, db, mode):
    class Rec(OpTimer):
        def launch(self, label, nbytes, fn, *a, **kw):
            self.records[label] = [1, 0.0, nbytes]
            fn(*a, **kw)
    pm.plan.timer = Rec()
    out = pm.call(db, mode)
    labels = set(pm.plan.timer.records)
    pm.plan.timer = None
    return out, labels


@pytest.mark.parametrize('depth,uv,k,n', [(256, 64, 1, 2), (256, 64, 2, 2), (256, 64, 4, 1), (256, 128, 3, 1), (1024, 256, 1, 1),
                                          (256, 64, 5, 1)])
def test_model_call_with_fused_ends_vs_oracle(depth, uv, k, n):
    # the front kernel (csrc/front4.hip, any k) also runs level 2's stride-2 convs; with the first-generation kernel
    # (plan.front_v4 = False) k = 5 takes the plain front and launches them separately -- both are checked for k = 5
    om, pm = make_pair(depth=depth, uv=uv, im=uv // 2, seed=depth + k)
    batch, nn = O.synth_batch(n, uv, uv, uv // 2, uv // 2, uv // 2, uv // 2, k=k, seed=20 + k)
    with torch.no_grad():
        o_pred_c, o_gt_c, _, o_vis = om.call(batch, 'vali', nn_list=nn)
    db = to_device_batch(batch, nn)
    (p_pred_c, p_gt_c, _, p_vis), labels = _labels(pm, db, 'vali')
    torch.cuda.synchronize()
    assert 'F.front' in labels and 'F.back' in labels and 'L0.stem' not in labels
    assert 'L2.o.s2' not in labels and 'L2.q.s2' not in labels and 'L2.o.s1' in labels
    assert rel_l2(p_vis['pred'].cpu(), o_vis['pred']) <= TOL
    assert rel_l2(p_pred_c.cpu(), o_pred_c) <= TOL and rel_l2(p_gt_c.cpu(), o_gt_c) <= 1e-6
    if k > 4:
        pm.plan.front_v4 = False
        pm.plan._drop_tapes()
        (g1_pred_c, _, _, g1_vis), labels = _labels(pm, db, 'vali')
        assert 'L2.o.s2' in labels and 'L2.q.s2' in labels and 'F.front' in labels
        assert rel_l2(g1_vis['pred'].cpu(), o_vis['pred']) <= TOL
        pm.plan.front_v4 = True
    pm.plan.fuse_ends = False
    (u_pred_c, _, _, u_vis), labels = _labels(pm, db, 'vali')
    assert 'L0.stem' in labels and 'F.front' not in labels
    assert rel_l2(p_vis['pred'].cpu(), u_vis['pred'].cpu()) <= 1e-5


def test_fused_vs_layer_by_layer_at_bench_shape():
    """BASELINE config 3 shape per frame (1024^2 UV, k = 4, 512^2 warp), 2 frames: the two plans agree."""
    import nlt_amd
    import bench
    from nlt_amd.models import get_model_class
    dev = torch.device('cuda', 0)
    model = get_model_class('nlt')(nlt_amd.make_config(uvh=1024, uvw=1024, imh=512, imw=512)).build(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    for v in model.register_trainable() or model.trainable_variables:
        if v.dim() == 1:
            v.data.uniform_(-0.1, 0.1, generator=g)
    batch = bench.synth_device_batch(2, 1024, 512, 4, dev, seed=5)
    a = model.call(batch, 'test')
    pred_a, cam_a = a[3]['pred'].clone(), a[0].clone()
    model.plan.fuse_ends = False
    b = model.call(batch, 'test')
    torch.cuda.synchronize()
    assert rel_l2(pred_a.cpu(), b[3]['pred'].cpu()) <= 1e-5
    assert rel_l2(cam_a.cpu(), b[0].cpu()) <= 1e-5
    assert not pred_a[:, 0, 0].any()


def test_front_weights_are_refolded_after_an_update():
    om, pm = make_pair(depth=256, uv=64, im=32, seed=3)
    batch, nn = O.synth_batch(1, 64, 64, 32, 32, 32, 32, k=2, seed=4)
    db = to_device_batch(batch, nn)
    first = pm.call(db, 'test')[3]['pred'].clone()
    om2 = O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32, seed=77)
    pm.load_weights(om2.numpy_weights())
    with torch.no_grad():
        ref = om2.call(batch, 'test', nn_list=nn)[3]['pred']
    second = pm.call(db, 'test')[3]['pred']
    assert rel_l2(second.cpu(), ref) <= TOL and rel_l2(first.cpu(), ref) > 1e-2


def test_fused_vs_layer_by_layer_at_2048():
    """BASELINE config 5's UV size (2048^2), one frame, k = 1, in fp32: the two forward plans agree and the frame's
    corner texel is the background sink.  (bf16 storage is not built; this pins the kernels' 32-bit index arithmetic
    and tiling at the largest listed resolution.)"""
    import nlt_amd
    import bench
    from nlt_amd.models import get_model_class
    dev = torch.device('cuda', 0)
    model = get_model_class('nlt')(nlt_amd.make_config(uvh=2048, uvw=2048, imh=512, imw=512)).build(dev)
    model.plan.autotune = False
    batch = bench.synth_device_batch(1, 2048, 512, 1, dev, seed=9)
    a = model.call(batch, 'test')
    pred_a = a[3]['pred'].clone()
    model.plan.fuse_ends = False
    b = model.call(batch, 'test')
    torch.cuda.synchronize()
    assert rel_l2(pred_a.cpu(), b[3]['pred'].cpu()) <= 1e-5
    assert not pred_a[:, 0, 0].any() and torch.isfinite(pred_a).all()


def test_hipgraph_replay_matches_eager_and_tracks_inputs():
    """model.use_graphs: third call with the same input tensors replays a captured hipGraph (two streams inside);
    results equal the eager launch bit for bit, follow in-place changes of the inputs, and a batch at other addresses
    falls back to eager launches."""
    import nlt_amd
    import bench
    from nlt_amd.models import get_model_class
    dev = torch.device('cuda', 0)
    model = get_model_class('nlt')(nlt_amd.make_config(uvh=256, uvw=256, imh=128, imw=128)).build(dev)
    batch = bench.synth_device_batch(2, 256, 128, 2, dev, seed=3)
    ref = model.call(batch, 'test')
    ref_pred, ref_cam = ref[3]['pred'].clone(), ref[0].clone()
    model.use_graphs = True
    for i in range(4):
        out = model.call(batch, 'test')
        torch.cuda.synchronize()
        assert torch.equal(out[3]['pred'], ref_pred) and torch.equal(out[0], ref_cam), i
    assert model._graph['graph'] is not None and model._graph['hits'] >= 2
    batch[1].mul_(0.5)                                            # base changes in place: same addresses -> replay sees it
    out = model.call(batch, 'test')
    model.use_graphs = False
    eager = model.call(batch, 'test')
    torch.cuda.synchronize()
    assert torch.equal(out[3]['pred'], eager[3]['pred']) and not torch.equal(out[3]['pred'], ref_pred)
    model.use_graphs = True
    other = bench.synth_device_batch(2, 256, 128, 2, dev, seed=4)
    o2 = model.call(other, 'test')                               # new addresses: eager again
    model.use_graphs = False
    e2 = model.call(other, 'test')
    torch.cuda.synchronize()
    assert torch.equal(o2[3]['pred'], e2[3]['pred'])
