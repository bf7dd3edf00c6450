"""CPU: the host orchestration of the product (fused RenderPlan: interleaved feature maps,
channel-slice outputs, dual-source virtual concat, bottleneck self-concat, obs_override; the
layer-wise Model._call; Model.call's warp/resize/blend sequence) driven through a TEST-ONLY
emulation of the C-ABI adapters (tests/fake_capi.py) and compared with the oracle.  The real
kernels are checked by the -m gpu tests; this guards the Python plumbing on machines without a GPU."""
import numpy as np
import pytest
import torch

import nlt_amd
from nlt_amd.models import get_model_class
from oracle import nlt_oracle as O
import fake_capi


def rel_l2(a, b):
    return float(np.linalg.norm(a.numpy().astype(np.float64) - b.numpy()) / np.linalg.norm(b.numpy()))


def make(depth, uv, im, **kw):
    om = O.OracleModel(depth=depth, uvh=uv, uvw=uv, imh=im, imw=im, seed=1, **kw)
    pm = get_model_class('nlt')(nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=im, imw=im, **kw))
    # CPU weights (the fake adapters take CPU tensors)
    for name in ('query', 'obs'):
        for layer, lw in zip(pm.net[name].layers, om.numpy_weights()[name]):
            convs = [layer] if hasattr(layer, 'set_weights') else [c for c, _ in layer.convs()]
            for c, (k, b) in zip(convs, lw):
                c.kernel, c.bias = torch.tensor(k), torch.tensor(b)
                c.cin = k.shape[3] if c.transpose else k.shape[2]
                c.built = True
    return om, pm


def cpu_batch(batch, nn):
    b = list(batch)
    b[8] = torch.stack([x[0] for x in nn], 1); b[9] = torch.stack([x[1] for x in nn], 1)
    return tuple(b)


@pytest.mark.parametrize('depth,uv,k', [(256, 64, 1), (256, 64, 3), (1024, 256, 1)])
def test_fused_plan_matches_oracle(monkeypatch, depth, uv, k):
    fake_capi.install(monkeypatch)
    om, pm = make(depth, uv, 32)
    batch, nn = O.synth_batch(1, uv, uv, 16, 16, 32, 32, k=k, seed=2)
    with torch.no_grad():
        ref_c, ref_gt, _, ref_vis = om.call(batch, 'train', nn_list=nn)
    got_c, got_gt, kw, vis = pm.call(cpu_batch(batch, nn), 'train', want_indices=True)
    assert kw == {}
    assert rel_l2(vis['pred'], ref_vis['pred']) < 1e-5
    assert rel_l2(got_c, ref_c) < 1e-5 and rel_l2(got_gt, ref_gt) < 1e-5
    assert got_c.shape == (1, 32, 32, 3)                     # warp res 16 -> image res 32 (resize path)


def test_layerwise_call_and_override_and_flags(monkeypatch):
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 64)
    batch, nn = O.synth_batch(2, 64, 64, 64, 64, 64, 64, k=2, seed=3)
    x = torch.cat((batch[1], batch[2], batch[3]), 3)
    with torch.no_grad():
        ref, feats = om._call(x, [r - b for b, r in nn], return_feats=True)
        got = pm._call(x, [r - b for b, r in nn])
        assert rel_l2(got, ref) < 1e-5
        override = [f.mean(0, keepdim=True) for f in feats]
        ref_o = om.call(batch, 'test', obs_override=[f.expand(2, -1, -1, -1) for f in override], nn_list=nn)[3]['pred']
    got_o = pm.call(cpu_batch(batch, nn), 'test', obs_override=override)[3]['pred']
    assert rel_l2(got_o, ref_o) < 1e-5
    w = torch.rand(2, 2)
    with torch.no_grad():
        ref_w = om._call(x, [r - b for b, r in nn], obs_weights=w)
    pred_w, _ = pm.plan.forward(batch[1], batch[2], batch[3], cpu_batch(batch, nn)[9], cpu_batch(batch, nn)[8],
                                obs_weights=w, skip_connect_base=False)
    ref_w = ref_w.clone(); ref_w[:, 0, 0, :] = 0
    assert rel_l2(pred_w, ref_w) < 1e-5


def test_no_obs_no_skip(monkeypatch):
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 64, use_obs=False, skip_connect_base=False)
    batch, nn = O.synth_batch(1, 64, 64, 64, 64, 64, 64, k=1, seed=4)
    with torch.no_grad():
        ref = om.call(batch, 'vali', nn_list=nn)[3]['pred']
    assert rel_l2(pm.call(cpu_batch(batch, nn), 'vali')[3]['pred'], ref) < 1e-5


def test_relu_activation_config_branch(monkeypatch):
    """act = relu (elements.py:69-70, ReLU(negative_slope=0)) runs through the same fused epilogues with slope 0,
    forward (fused and layer-by-layer plans) and in the train step's derivative masks."""
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 32, act='relu')
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=12)
    with torch.no_grad():
        ref = om.call(batch, 'vali', nn_list=nn)[3]['pred']
    assert rel_l2(pm.call(cpu_batch(batch, nn), 'vali')[3]['pred'], ref) < 1e-5
    pm.plan.fuse_ends = False
    assert rel_l2(pm.call(cpu_batch(batch, nn), 'vali')[3]['pred'], ref) < 1e-5
    # elu (elements.py:74-75) is a stand-alone layer: the model takes the layer-by-layer path (tests/test_host_generic.py)
    assert get_model_class('nlt')(nlt_amd.make_config(depth=256, uvh=64, uvw=64, imh=32, imw=32, act='elu')).generic
    assert not pm.generic


def test_op_labels_and_algorithmic_bytes(monkeypatch):
    """Per-launch algorithmic bytes sum to SURVEY 8d's per-texel figure (961.5 B k=1; 1755.75 B k=4)."""
    fake_capi.install(monkeypatch)
    from nlt_amd.engine import OpTimer

    class Rec(OpTimer):
        def launch(self, label, nbytes, fn, *a, **kw):
            self.records[label] = [1, 0.0, nbytes]
            fn(*a, **kw)
    for k, per_texel in ((1, 961.5), (4, 1755.75)):
        for fused in (False, True):
            om, pm = make(256, 64, 64)
            pm.plan.fuse_ends = fused
            pm.plan.timer = Rec()
            batch, nn = O.synth_batch(1, 64, 64, 64, 64, 64, 64, k=k, seed=5)
            pm.call(cpu_batch(batch, nn), 'test')
            total = sum(r[2] for r in pm.plan.timer.records.values())
            # the explicit obs-mean launches come on top of the layer-wise accounting; unfused, so do the stem's
            # second raw obs input + mean write and the head's base read.  Everything else must match exactly.
            extra = sum(r[2] for l, r in pm.plan.timer.records.items() if l.endswith('.o.mean'))
            if not fused:
                extra += 4 * 64 * 64 * (3 * k + 16) + 4 * 64 * 64 * 3
            assert abs((total - extra) / (64 * 64) - per_texel) < 1e-6, (k, fused, (total - extra) / 4096)
            # fused: front (L0, L1 and level 2's two stride-2 convs) + 5 levels x 5 - 2 + 3 decoder blocks x 2 + the two
            # single-launch blocks with 16 / 8 output channels (csrc/dec_block.hip) + back
            # (k = 1, r06: no mean launches -- the observation features are written straight into the interleaved map)
            want_fused = 1 + 5 * (5 if k > 1 else 4) - 2 + 3 * 2 + 2 + 1
            assert len(pm.plan.timer.records) == (want_fused if fused else 1 + 6 * 5 + 6 * 2 + 1)
            assert not (fused and k == 1 and any(l.endswith('.o.mean') for l in pm.plan.timer.records))
            assert ('F.front' in pm.plan.timer.records) == fused and ('L0.stem' in pm.plan.timer.records) != fused


def test_fused_ends_match_unfused_and_are_skipped_when_they_must_be(monkeypatch):
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 32)
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=8)
    cb = cpu_batch(batch, nn)
    with torch.no_grad():
        ref = om.call(batch, 'test', nn_list=nn)[3]['pred']
    a = pm.call(cb, 'test')[3]['pred']
    pm.plan.fuse_ends = False
    b = pm.call(cb, 'test')[3]['pred']
    assert rel_l2(a, ref) < 1e-5 and rel_l2(b, ref) < 1e-5
    pm.plan.fuse_ends = True
    bufs = pm.plan._buffers(2, 2, 64, 64, cb[1].device)
    assert pm.plan.can_fuse(bufs, None, None)
    assert not pm.plan.can_fuse(bufs, torch.ones(2, 2), None)       # obs_weights -> layer-by-layer kernels
    assert not pm.plan.can_fuse(bufs, None, [None])                  # obs_override (nlt_test.py) -> layer-by-layer
    pm.plan.fuse_ends = False
    assert not pm.plan.can_fuse(bufs, None, None)
    pm.plan.fuse_ends = True
    pm.plan.front_v4 = False
    assert not pm.plan.can_fuse(pm.plan._buffers(1, 15, 64, 64, cb[1].device), None, None)   # k = 15: LDS of the first-generation front kernel
    pm.plan.front_v4 = True
    assert pm.plan.can_fuse(pm.plan._buffers(1, 15, 64, 64, cb[1].device), None, None)       # csrc/front4.hip streams the observations


@pytest.mark.parametrize('fused', [False, True])
def test_lds_tiled_encoder_launches_fold_the_observation_mean(monkeypatch, fused):
    """Plan with every eligible encoder conv routed to csrc/conv_tile.hip: same result, no '.o.mean' launch
    where the tiled kernel produced the mean itself."""
    fake_capi.install(monkeypatch)
    from nlt_amd.engine import OpTimer

    class Rec(OpTimer):
        def launch(self, label, nbytes, fn, *a, **kw):
            self.records[label] = [1, 0.0, nbytes]
            fn(*a, **kw)
    om, pm = make(256, 64, 32)
    pm.plan.fuse_ends = fused
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=3, seed=9)
    with torch.no_grad():
        ref = om.call(batch, 'test', nn_list=nn)[3]['pred']
    labels = ['L%d.%s.%s' % (l, p, s) for l in range(1, 7) for p in 'qo' for s in ('s1', 's2')]
    pm.plan.lds_hints = {lab: 32 for lab in labels}
    pm.plan.timer = Rec()
    got = pm.call(cpu_batch(batch, nn), 'test')[3]['pred']
    assert rel_l2(got, ref) < 1e-5
    rec = pm.plan.timer.records
    means = sorted(l for l in rec if l.endswith('.o.mean'))
    assert means == ([] if fused else ['L1.o.mean'])            # level 1 has 16 output channels: not eligible
    assert pm.plan._ran_lds >= {'L3.q.s2', 'L3.o.s2', 'L2.q.s1', 'L2.o.s1', 'L6.o.s1'}
    assert ('L2.o.s2' in pm.plan._ran_lds) != fused             # fused: level 2's stride-2 convs ran inside the front kernel
    assert 'L1.q.s2' not in pm.plan._ran_lds


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('hint', [32, 64, 256 + 32, 256 + 64])
def test_winograd_launches_of_the_plan(monkeypatch, fused, hint):
    """Plan with every eligible stride-1 k2 launch given to the Winograd kernels: same result; the observation mean is folded
    into the launch (no '.o.mean' launch) unless the hint says 'unfolded' (+256: observations as frames, the mean in its own
    launch); the expanding blocks' transposed stride-1 convs go there too."""
    fake_capi.install(monkeypatch)
    from nlt_amd.engine import OpTimer

    class Rec(OpTimer):
        def launch(self, label, nbytes, fn, *a, **kw):
            self.records[label] = [1, 0.0, nbytes]
            fn(*a, **kw)
    om, pm = make(256, 64, 32)
    pm.plan.fuse_ends = fused
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=3, seed=9)
    with torch.no_grad():
        ref = om.call(batch, 'test', nn_list=nn)[3]['pred']
    labels = ['L%d.%s.%s' % (l, p, s) for l in range(1, 13) for p in 'qo' for s in ('s1', 's2')]
    pm.plan.wino_hints = {lab: hint for lab in labels}
    pm.plan.timer = Rec()
    got = pm.call(cpu_batch(batch, nn), 'test')[3]['pred']
    assert rel_l2(got, ref) < 1e-5
    rec = pm.plan.timer.records
    means = sorted(l for l in rec if l.endswith('.o.mean'))
    ran = pm.plan._ran_wino
    assert not any(l.endswith('.s2') for l in ran)
    eligible = [l for l in range(2, 7) if hint & 255 == 32 or l >= 3]       # level 2 has 32 channels: not a multiple of 64
    assert ran >= {'L%d.o.s1' % l for l in eligible} | {'L%d.q.s1' % l for l in eligible}
    assert ('L7.q.s1' in ran) and ('L9.q.s1' in ran) == (hint & 255 == 32)   # expanding blocks: 128 / 64 / 32 channels
    folded = hint < 256
    want = [] if fused else ['L1.o.mean']
    if not folded:
        want = sorted(want + ['L%d.o.mean' % l for l in eligible])
    if hint & 255 == 64:
        want = sorted(set(want) | {'L2.o.mean'})                              # level 2 stays on the register-tiled kernel + its mean
    assert means == want, (means, want)


@pytest.mark.parametrize('hint', [1, 2])
def test_narrow_level_launches_of_the_plan(monkeypatch, hint):
    """Level 2's stride-1 convs (32 -> 32) given to csrc/conv_c32.hip: same result; hint 1 folds the observation mean into the
    launch, hint 2 runs the observations as frames and keeps the '.o.mean' launch."""
    fake_capi.install(monkeypatch)
    from nlt_amd.engine import OpTimer

    class Rec(OpTimer):
        def launch(self, label, nbytes, fn, *a, **kw):
            self.records[label] = [1, 0.0, nbytes]
            fn(*a, **kw)
    om, pm = make(256, 64, 32)
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=3, seed=9)
    with torch.no_grad():
        ref = om.call(batch, 'test', nn_list=nn)[3]['pred']
    labels = ['L%d.%s.%s' % (l, p, s) for l in range(1, 13) for p in 'qo' for s in ('s1', 's2')]
    pm.plan.c32_hints = {lab: hint for lab in labels}
    pm.plan.timer = Rec()
    got = pm.call(cpu_batch(batch, nn), 'test')[3]['pred']
    assert rel_l2(got, ref) < 1e-5
    assert pm.plan._ran_c32 == {'L2.q.s1', 'L2.o.s1'}
    assert ('L2.o.mean' in pm.plan.timer.records) == (hint == 2)


def test_nlt_test_orchestration_extract_feat_and_infer(monkeypatch):
    """nlt/nlt_test.py:78-127: observation features averaged over all training frames, then used as obs_override;
    the observation convs are not launched at all during inference."""
    fake_capi.install(monkeypatch)
    from nlt_amd import nlt_test
    from nlt_amd.engine import OpTimer

    class Rec(OpTimer):
        def launch(self, label, nbytes, fn, *a, **kw):
            self.records[label] = [1, 0.0, nbytes]
            fn(*a, **kw)
    om, pm = make(256, 64, 32)
    train = [O.synth_batch(n, 64, 64, 32, 32, 32, 32, k=1, seed=40 + n) for n in (2, 3)]     # unequal batch sizes
    test_b, test_nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=50)
    with torch.no_grad():
        feats = []
        for b, _ in train:
            _, f = om._call(torch.cat((b[1], b[2], b[3]), 3), [b[5] - b[1]], return_feats=True)   # x = rgb - base
            feats.append(f)
        ref_agg = [torch.cat([f[l] for f in feats], 0).mean(0, keepdim=True) for l in range(len(feats[0]))]
        ref = om.call(test_b, 'test', obs_override=[f.expand(2, -1, -1, -1) for f in ref_agg], nn_list=test_nn)[3]['pred']
    agg = nlt_test.extract_feat(pm, [cpu_batch(b, nn) for b, nn in train])
    assert len(agg) == len(ref_agg) and all(a.shape == r.shape and rel_l2(a, r) < 1e-5 for a, r in zip(agg, ref_agg))
    assert len(nlt_test.extract_feat(pm, [cpu_batch(b, nn) for b, nn in train], n_obs_batches=1)) == len(ref_agg)
    pm.plan.timer = Rec()
    out = nlt_test.infer(pm, [cpu_batch(test_b, test_nn)], agg)
    assert rel_l2(out[0]['pred'], ref) < 1e-5
    assert not any('.o.' in l or l == 'L0.stem' for l in pm.plan.timer.records)          # no observation launches
    with pytest.raises(ValueError):
        nlt_test.extract_feat(pm, [])


@pytest.mark.parametrize('depth,uv,n', [(256, 64, 2), (1024, 256, 1), (64, 32, 3)])
def test_fused_override_plan_matches_oracle_and_the_general_plan(monkeypatch, depth, uv, n):
    """engine_infer.py: Model.call(obs_override = one map per level) on the fused query-only plan -- override maps, derived
    query-row convs, bottleneck self-concat folded into one kernel -- against the oracle's `_call(obs_override=...)` and against
    the layer-by-layer plan; a per-frame override (a real [N, ...] tensor) must keep taking the general plan."""
    fake_capi.install(monkeypatch)
    from nlt_amd.engine import OpTimer

    class Rec(OpTimer):
        def launch(self, label, nbytes, fn, *a, **kw):
            self.records[label] = [1, 0.0, nbytes]
            fn(*a, **kw)
    om, pm = make(depth, uv, 32)
    batch, nn = O.synth_batch(n, uv, uv, 32, 32, 32, 32, k=1, seed=60)
    x = torch.cat((batch[1], batch[2], batch[3]), 3)
    with torch.no_grad():
        _, feats = om._call(x, [r - b for b, r in nn], return_feats=True)
        agg = [f.mean(0, keepdim=True) * 1.7 + 0.05 for f in feats]
        ref = om.call(batch, 'test', obs_override=[f.expand(n, -1, -1, -1) for f in agg], nn_list=nn)
    pm.plan.timer = Rec()
    got = pm.call(cpu_batch(batch, nn), 'test', obs_override=agg)
    recs = set(pm.plan.timer.records)
    assert 'F.front' in recs and 'F.back' in recs and 'V3.q.s2' in recs and not any('.o.' in l or l == 'L0.stem' for l in recs)
    assert rel_l2(got[3]['pred'], ref[3]['pred']) < 1e-5 and rel_l2(got[0], ref[0]) < 1e-5
    # an expand()ed view is the same thing; a materialised per-frame override is not
    pm.plan.timer = Rec()
    got2 = pm.call(cpu_batch(batch, nn), 'test', obs_override=[f.expand(n, -1, -1, -1) for f in agg])
    assert torch.equal(got2[3]['pred'], got[3]['pred']) or n == 1
    # ... and so is what the reference's own call site builds: tf.tile(x, (bs, 1, 1, 1)) (nlt/nlt_test.py:83-86), a materialised copy per frame
    pm.plan.timer = Rec()
    tiled = [f.repeat(n, 1, 1, 1) for f in agg]
    got2t = pm.call(cpu_batch(batch, nn), 'test', obs_override=tiled)
    assert 'F.front' in pm.plan.timer.records and torch.equal(got2t[3]['pred'], got[3]['pred'])
    pm.plan.timer = Rec()
    per_frame = [f.repeat(n, 1, 1, 1) * torch.linspace(1.0, 1.5, n).view(n, 1, 1, 1) for f in agg]
    with torch.no_grad():
        ref3 = om.call(batch, 'test', obs_override=per_frame, nn_list=nn)
    got3 = pm.call(cpu_batch(batch, nn), 'test', obs_override=per_frame)
    assert 'F.front' not in pm.plan.timer.records or n == 1
    assert rel_l2(got3[3]['pred'], ref3[3]['pred']) < 1e-5
    # the general plan on the shared maps (NLT_FUSED_OVERRIDE=0) agrees with the fused one
    pm.plan.fuse_override = False
    pm.plan.timer = Rec()
    gen = pm.call(cpu_batch(batch, nn), 'test', obs_override=agg)
    assert 'F.front' not in pm.plan.timer.records
    assert rel_l2(gen[3]['pred'], got[3]['pred']) < 1e-5
    # new weights -> new maps
    pm.plan.fuse_override = True
    pm.plan.timer = None
    serial = pm.plan._ovr['serial']
    head = pm.net['query'].layers[-1]
    head.kernel = head.kernel * 1.25
    got4 = pm.call(cpu_batch(batch, nn), 'test', obs_override=agg)
    assert pm.plan._ovr['serial'] != serial
    pm.plan.fuse_override = False
    gen4 = pm.call(cpu_batch(batch, nn), 'test', obs_override=agg)
    assert rel_l2(got4[3]['pred'], gen4[3]['pred']) < 1e-5 and rel_l2(got4[3]['pred'], got[3]['pred']) > 1e-3


def test_render_pipeline_lane_bookkeeping_on_the_host(monkeypatch):
    """pipeline.RenderPipeline without a GPU (lanes are a launch-scheduling matter; on CPU tensors every lane just calls):
    batches go round-robin to lanes, a lane is the model's render state only (own plan, shared nets / weights), the
    lanes copy lane 0's tile choices, results come back in order and equal Model.call's."""
    fake_capi.install(monkeypatch)
    from nlt_amd import nlt_test
    from nlt_amd.pipeline import RenderPipeline
    om, pm = make(256, 64, 32)
    data = [O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=80 + i) for i in range(5)]
    batches = [cpu_batch(b, nn) for b, nn in data]
    ref = [pm.call(b, 'test')[3]['pred'] for b in batches]
    pm.plan.tile_hints['L3.q.s1'] = 18                          # (as if lane 0's plan-time trials had chosen it)
    pm.plan.tuned = {'L3.q.s1': [(1.0, 'tile', 18)]}
    pipe = RenderPipeline(pm, lanes=3)
    seen = []
    pipe.render(batches, 'test', on_batch=lambda i, r: seen.append((i, r[3]['pred'])))
    assert [i for i, _ in seen] == list(range(5))
    assert all(torch.equal(a, b) for a, (_, b) in zip(ref, seen))
    l1, l2 = pipe._lanes[1], pipe._lanes[2]
    assert l1 is not pm and l1.plan is not pm.plan and l2.plan is not l1.plan
    assert l1.net is pm.net and l1.plan.q is pm.plan.q and l1.plan.o is pm.plan.o
    assert l1.plan.tile_hints == pm.plan.tile_hints and l1.plan.autotune is False
    assert pm.plan.generation == 5 + 2 and l1.plan.generation == 2 and l2.plan.generation == 1   # batches 0,3 | 1,4 | 2 (+ the 5 reference calls)
    out = nlt_test.infer(pm, batches[:3], None, lanes=2)
    assert len(out) == 3 and torch.equal(out[2]['pred'], ref[2])
    with pytest.raises(ValueError):
        RenderPipeline(pm, 0)


def test_fused_override_plan_on_a_store_resident_batch(monkeypatch):
    """RenderPlan.forward(resident=ResidentTexels, obs_override=maps): the override plan with its front launch reading the uint8
    capture store by frame id -- same texels as the plan on the float batch `_load_data` assembles (nlt/datasets/nlt.py:131-136)."""
    fake_capi.install(monkeypatch)
    from nlt_amd import _capi as C
    from nlt_amd.datasets.nlt import ResidentTexels
    uv, n = 64, 2
    om, pm = make(256, uv, 32)
    g = torch.Generator().manual_seed(3)
    U = lambda *s: torch.randint(0, 256, s, generator=g, dtype=torch.uint8)
    store = {'diffuse': U(5, uv, uv, 3), 'rgb': U(5, uv, uv, 3), 'cvis': U(5, uv, uv), 'lvis': U(5, uv, uv),
             'uv2cam': torch.rand(5, 32, 32, 2, generator=g).half()}
    ids, nn_ids = torch.tensor([3, 1], dtype=torch.int32), torch.tensor([[0], [2]], dtype=torch.int32)
    res = ResidentTexels(store, ids, nn_ids, test_mode=True)
    fl = res.materialize()
    batch, nn = O.synth_batch(n, uv, uv, 32, 32, 32, 32, k=1, seed=61)
    with torch.no_grad():
        _, feats = om._call(torch.cat((batch[1], batch[2], batch[3]), 3), [r - b for b, r in nn], return_feats=True)
    agg = [f.mean(0, keepdim=True) for f in feats]
    plan = pm.plan
    assert plan.resident_override_ok(res, agg, 0.3)
    a, _ = plan.forward(fl['base'], fl['cvis'], fl['lvis'], fl['nn_rgb'], fl['nn_base'], obs_override=agg, inference=True)
    a = a.clone()
    b, _ = plan.forward(None, None, None, None, None, obs_override=agg, inference=True, resident=res)
    assert plan._ovr is not None and rel_l2(b, a) < 1e-6
    with torch.no_grad():
        ref = om._call(torch.cat((fl['base'], fl['cvis'], fl['lvis']), 3), [fl['nn_rgb'][:, 0] - fl['nn_base'][:, 0]],
                       obs_override=[f.expand(n, -1, -1, -1) for f in agg])
    want = ref + fl['base']
    want[:, 0, 0, :] = 0
    assert rel_l2(b, want) < 1e-5
    # an override whose frames differ cannot take the store-resident path (copies of one map can: the reference's tf.tile)
    assert plan.resident_override_ok(res, [f.repeat(n, 1, 1, 1) for f in agg], 0.3)
    per_frame = [f.repeat(n, 1, 1, 1) * torch.linspace(1.0, 1.5, n).view(n, 1, 1, 1) for f in agg]
    assert not plan.resident_override_ok(res, per_frame, 0.3)
    with pytest.raises(C.NLTError):
        plan.forward(None, None, None, None, None, obs_override=per_frame, inference=True, resident=res)
