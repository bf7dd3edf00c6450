"""-m gpu: full train step on HIP kernels (forward plan, losses, backward plan, flat gradient
bucket, fused Adam-AMSGrad) vs the oracle's torch-CPU autograd train step (nlt/trainvali.py:272-281)."""
import numpy as np
import pytest
import torch

import nlt_amd
from nlt_amd import trainvali
from oracle import nlt_oracle as O
from gpu_util import make_pair, to_device_batch

pytestmark = pytest.mark.gpu


def flat_oracle_grads(pm, grads):
    out = torch.zeros_like(pm.flat_grads)
    it = iter(grads)
    for c in pm._conv_layers():
        for name in ('dkernel', 'dbias'):
            g = next(it)
            off = (getattr(c, name).data_ptr() - pm.flat_grads.data_ptr()) // 4
            out[off: off + g.numel()] = g.reshape(-1).to(out.device)
    return out


def per_tensor_worst(pm, grads):
    """(largest rel-L2 error over every kernel / bias gradient, its index) -- a flat-bucket norm is dominated by the big
    deep kernels and cannot see a wrong 16-float bias gradient."""
    it = iter(grads)
    worst = (0.0, None)
    for i, c in enumerate(pm._conv_layers()):
        for name in ('dkernel', 'dbias'):
            g = next(it)
            e = float((getattr(c, name).detach().cpu() - g).norm() / (g.norm() + 1e-30))
            worst = max(worst, (e, '%d.%s' % (i, name)))
    return worst


# fp32 HIP gradients vs the fp32 torch-CPU oracle: both sides carry fp32 accumulation error (different summation
# orders); the float64 comparison at BASELINE config 4's size is in tests/test_gpu_baseline_sizes.py
# (single tensors: LeakyReLU-derivative flips of texels whose pre-activation is within fp32 rounding of zero move a small
# deep-layer gradient by up to ~2e-3 between ANY two fp32 evaluation orders -- the fp32 and float64 torch-CPU oracles
# differ by 1e-3 among themselves; the kink-free alpha = 1 comparison there holds every tensor to 5e-5)
FLAT_TOL, TENSOR_TOL = 1e-4, 5e-3


# ---- the multi-rank code path first: a tolerance-sensitive comparison further down can never keep these from running (-x)
def test_rccl_single_rank_group_runs_the_gradient_all_reduce():
    """One-rank `nccl` (= RCCL) process group on the GPU box: the flat gradient bucket goes through the real
    collective library (the world-size-2 logic is covered on CPU by tests/test_dist_gloo.py)."""
    import os
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    try:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    except Exception as e:                                    # no usable RCCL / rendezvous on this box: not a kernel failure
        pytest.skip("RCCL process group could not be created: %r" % (e,))
    try:
        g = torch.randn(3368072, device='cuda')
        ref = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        t = torch.tensor([1.5], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(g, ref) and float(t) == 1.5
    finally:
        dist.destroy_process_group()


def test_rccl_two_bucket_overlapped_step_equals_the_single_rank_step(monkeypatch):
    """The multi-rank code path of trainvali.distributed_train_step (gradient all-reduce in two ranges, the first issued
    as an async RCCL collective from inside the backward plan on the weight-gradient stream, launch tape replays included)
    on a ONE-rank `nccl` group with the world size patched to 2: the collectives are identities, so weights and losses must
    track the plain single-rank step exactly (up to the float atomics of the warp adjoint)."""
    import os
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29537')
    try:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    except Exception as e:
        pytest.skip("RCCL process group could not be created: %r" % (e,))
    try:
        batches = [to_device_batch(*O.synth_batch(2, 128, 128, 64, 64, 64, 64, k=1, seed=80 + i)) for i in range(2)]
        res = []
        tuning = None
        for multi in (False, True):
            _, pm = make_pair(depth=256, uv=128, im=64, loss='l2', seed=12)
            pm.build('cuda')
            if tuning is not None:
                pm.plan.import_tuning(tuning)
            opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
            monkeypatch.setattr(trainvali, '_world', (lambda g: 2) if multi else (lambda g: 1))
            losses = [float(trainvali.distributed_train_step(pm, batches[i % 2], opt, 2)[0]) for i in range(6)]
            torch.cuda.synchronize()
            tuning = pm.plan.export_tuning()
            res.append((losses, pm.flat_params.detach().clone(), pm.plan.tape_replays))
        (l0, p0, _), (l1, p1, replays) = res
        np.testing.assert_allclose(l1, l0, rtol=1e-4)
        assert float((p0 - p1).abs().max()) < 1e-4
        assert replays > 0                                       # the hook also fires from replayed tapes
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('wgrad_streams', [1, 2])
def test_overlapped_bucket_collective_runs_on_the_weight_gradient_stream_in_replayed_steps_too(monkeypatch, wgrad_streams):
    """The first-range gradient collective is started by a hook recorded INSIDE the backward plan.  It has to be queued on
    the stream the expanding blocks' weight-gradient launches run on (the side stream) -- also when the step is a launch-tape
    replay, where no plan is being issued and the side-stream cursor does not exist.  A stand-in collective that is NOT an
    identity (in-place x2 on the stream it is called on, completion = an event on that stream) makes a misplaced hook
    visible: on the main stream it would double half-accumulated sums.  overlap=True must match overlap=False, and every
    first-range call of an overlapped step must sit on the side stream.  wgrad_streams = 2: the weight gradients dealt to two
    side streams alternately (NLT_BWD_STREAMS=2); the hook's stream then waits for the other one first."""
    import torch.distributed as dist

    class Work:
        def __init__(self):
            self.ev = torch.cuda.Event()
            self.ev.record(torch.cuda.current_stream())

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    calls = []

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        calls.append((t.numel(), torch.cuda.current_stream().cuda_stream))
        if t.numel() > 1:
            t.mul_(2.0)
        return Work()

    monkeypatch.setattr(dist, 'all_reduce', fake_all_reduce)
    monkeypatch.setattr(trainvali, '_world', lambda g: 2)
    batches = [to_device_batch(*O.synth_batch(2, 128, 128, 64, 64, 64, 64, k=1, seed=90 + i)) for i in range(2)]
    res = []
    tuning = None
    for overlap in (False, True):
        _, pm = make_pair(depth=256, uv=128, im=64, loss='l2', seed=13)
        pm.build('cuda')
        pm.plan.bwd_streams = wgrad_streams
        if tuning is not None:
            pm.plan.import_tuning(tuning)        # both legs issue the SAME kernels: what differs is where the collective sits
        opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
        del calls[:]
        grads = []
        for i in range(8):
            trainvali.distributed_train_step(pm, batches[i % 2], opt, 2, overlap=overlap)
            grads.append(pm.flat_grads.clone())
        torch.cuda.synchronize()
        tuning = pm.plan.export_tuning()
        res.append((grads, pm.flat_params.detach().clone(), list(calls), pm.plan.tape_replays, pm.plan._bside, pm.bucket_split))
    (g0, p0, c0, _, _, _), (g1, p1, c1, replays, bside, split) = res
    assert replays > 0 and bside is not None
    main = torch.cuda.current_stream().cuda_stream
    side = bside[0].cuda_stream
    assert side != main and (bside[3] is not None) == (wgrad_streams == 2)
    first = [st for n, st in c1 if n == split]
    assert len(first) == 8 and all(st == side for st in first), (first, side, main)      # replayed steps included
    assert all(st == main for n, st in c0)                                              # overlap=False: after the backward
    assert len(c0) == len(c1) == 8 * 4                                                  # three gradient ranges + the loss
    mid = [st for n, st in c1 if n == pm.bucket_ranges[2] - pm.bucket_ranges[1]]
    assert len(mid) == 8 and all(st == side for st in mid)                              # the second range: from inside the backward too
    # A misplaced hook doubles half-accumulated sums: an O(1) error.  What two CORRECT runs differ by is the float-atomic
    # noise of the resampler adjoint (~1e-7 of the gradient at step 0), which Adam's m / sqrt(v) then carries into the
    # weights of the later steps: step 0 is bounded tightly, the replayed steps by a drift-aware bound.
    rel = [float((a - b).norm() / a.norm()) for a, b in zip(g0, g1)]
    assert rel[0] < 5e-6, rel
    assert max(rel) < 1e-3, rel
    assert float((p0 - p1).abs().max()) < 1e-3


def _two_rank_worker(rank, port, outdir):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    import nlt_amd
    from nlt_amd import trainvali
    from oracle import nlt_oracle as O
    from gpu_util import make_pair, to_device_batch
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    out = {}
    batch, nn = O.synth_batch(4, 128, 128, 64, 64, 64, 64, k=1, seed=77)             # the GLOBAL batch; this rank's half below
    sl = slice(2 * rank, 2 * rank + 2)
    shard = tuple(t[sl] if torch.is_tensor(t) else t for t in batch)
    db = to_device_batch(shard, [(b[sl], r[sl]) for b, r in nn])
    tuning = None
    for overlap in (False, True):
        _, pm = make_pair(depth=256, uv=128, im=64, loss='l2', seed=21)
        pm.build('cuda')
        if tuning is not None:
            pm.plan.import_tuning(tuning)
        opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
        losses = [float(trainvali.distributed_train_step(pm, db, opt, 4, overlap=overlap)[0]) for _ in range(6)]
        torch.cuda.synchronize()
        tuning = pm.plan.export_tuning()
        out[overlap] = (losses, pm.flat_params.detach().cpu().clone(), pm.plan.tape_replays)
    torch.save(out, os.path.join(outdir, 'r%d.pt' % rank))
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_overlapped_bucket_equals_serial_and_ranks_stay_identical():
    """Two real processes (one rank each, both on cuda:0, `gloo` carrying the CUDA tensors -- RCCL refuses two ranks on one
    device) run six data-parallel train steps on the HIP kernels: launch-tape replays, the first gradient range all-reduced
    from inside the backward plan on the weight-gradient stream.  The collective is a REAL sum here (each rank holds other
    frames), so a hook fired on the wrong stream -- reading half-accumulated weight gradients -- shows as rank divergence or
    as a difference between overlap=True and overlap=False.  Ranks must hold bit-identical weights; the two modes agree to
    the float-atomic noise of the warp adjoint."""
    import os
    import tempfile
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as td:
        port = 29700 + os.getpid() % 200
        mp.spawn(_two_rank_worker, args=(port, td), nprocs=2, join=True)
        r = [torch.load(os.path.join(td, 'r%d.pt' % i)) for i in range(2)]
    for overlap in (False, True):
        assert r[0][overlap][0] == r[1][overlap][0]                                   # the all-reduced loss
        assert torch.equal(r[0][overlap][1], r[1][overlap][1])                        # bit-identical weights on both ranks
    assert r[0][True][2] > 0                                                          # replayed steps were part of it
    np.testing.assert_allclose(r[0][True][0], r[0][False][0], rtol=1e-4)
    assert float((r[0][True][1] - r[0][False][1]).abs().max()) < 1e-4


@pytest.mark.parametrize('loss,k,uv,cam,im', [('l2', 1, 64, 64, 64), ('l2', 4, 128, 64, 64), ('barron', 2, 64, 32, 32),
                                              ('barron,5e-1l2', 1, 64, 32, 48)])
def test_train_step_matches_oracle(loss, k, uv, cam, im):
    om, pm = make_pair(depth=256, uv=uv, im=im, loss=loss, seed=k)
    pm.build('cuda')
    batch, nn = O.synth_batch(2, uv, uv, cam, cam, im, im, k=k, seed=20 + k)
    db = to_device_batch(batch, nn)
    opt_o = O.KerasAdamAMSGrad(om.parameters(), 1e-3)
    opt_p = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
    for step in range(3):
        lo, go = O.train_step(om, opt_o, batch, global_bs=2, nn_list=nn)
        lp, _ = trainvali.distributed_train_step(pm, db, opt_p, global_bs=2)
        torch.cuda.synchronize()
        assert abs(float(lp) - float(lo)) <= 2e-5 * max(1.0, abs(float(lo))), (step, float(lp), float(lo))
        ref = flat_oracle_grads(pm, go)
        rel = float((pm.flat_params.grad - ref).norm() / ref.norm())
        assert rel < FLAT_TOL, (step, rel)
        worst = per_tensor_worst(pm, go)
        assert worst[0] < TENSOR_TOL, (step, worst)
    # three Adam steps later the weights still track the oracle (lr 1e-3: each step moves ~1e-3)
    worst = max(float((po.detach() - c.kernel.cpu()).abs().max()) for po, c in zip(om.parameters()[::2], pm._conv_layers()))
    assert worst < 2e-4, worst


def test_loss_decreases_and_vali_step():
    om, pm = make_pair(depth=256, uv=64, im=64, loss='l2', seed=3)
    pm.build('cuda')
    batch, nn = O.synth_batch(4, 64, 64, 64, 64, 64, 64, k=1, seed=31)
    db = to_device_batch(batch, nn)
    opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
    losses = [float(trainvali.distributed_train_step(pm, db, opt, 4)[0]) for _ in range(8)]
    assert losses[-1] < losses[0]
    lv, vis = trainvali.distributed_vali_step(pm, db, 4)
    assert np.isfinite(float(lv)) and not vis['pred_camspc'].requires_grad


def test_graphed_train_step_replays_the_same_step_as_the_eager_one():
    """trainvali.GraphedTrainStep: forward + loss + backward as one hipGraph (new batches copied into its static inputs)
    against the kernel-by-kernel step from the same initial weights; differences = float atomics in the warp scatter."""
    batches = [to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=40 + i)) for i in range(3)]
    results = []
    tuning = None
    for graphed in (False, True):
        _, pm = make_pair(depth=256, uv=64, im=32, loss='l2', seed=9)
        pm.build('cuda')
        if tuning is not None:
            pm.plan.import_tuning(tuning)        # same kernels, same summation orders in both legs
        opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
        step = trainvali.GraphedTrainStep(pm, opt, 2, warmup=1) if graphed else None
        losses = []
        for it in range(6):
            b = batches[it % 3]
            loss, vis = step(b) if graphed else trainvali.distributed_train_step(pm, b, opt, 2)
            losses.append(float(loss))
        if graphed:
            assert step.failed is None and step.graph is not None and step.static_batch() is not None
        tuning = pm.plan.export_tuning()
        results.append((losses, pm.flat_params.detach().clone()))
    (l0, p0), (l1, p1) = results
    np.testing.assert_allclose(l1, l0, rtol=1e-4)                # (measured ~1e-6: atomics noise carried through six Adam steps)
    assert float((p0 - p1).abs().max()) < 1e-4


@pytest.mark.parametrize('loss', ['l2', 'barron'])
def test_full_size_gradient_agrees_with_directional_finite_differences(loss):
    """BASELINE config 4's per-GPU shape (1024^2 UV, 512^2 camera; 2 frames here), a size-independent property check
    BESIDE the direct oracle comparison of tests/test_gpu_baseline_sizes.py: the whole fused training path (front/back
    forward with kept activations, warp adjoint, fused backward ends, narrow / tiled weight gradients, flat bucket)
    must satisfy (L(w + e d) - L(w - e d)) / 2e = <grad L, d> along the gradient and along its parts owned by each
    fused backward end."""
    _, pm = make_pair(depth=256, uv=1024, im=512, loss=loss, seed=21)
    pm.build('cuda')
    db = to_device_batch(*O.synth_batch(2, 1024, 1024, 512, 512, 512, 512, k=1, seed=77))

    def value():
        with torch.no_grad():
            pred, gt, kw, _ = pm(db, mode='vali')
            return float(pm.compute_loss(pred, gt, keep_batch=True).double().sum() / 2)

    pred, gt, kw, _ = pm(db, mode='train')
    lv = pm.compute_loss(pred, gt, keep_batch=True).sum() / 2
    pm.flat_params.grad = None
    lv.backward()
    g = pm.flat_params.grad.detach().clone().double()
    assert abs(value() - float(lv.detach())) <= 1e-5 * abs(float(lv.detach()))       # train and vali forwards agree
    # directions: the whole gradient, and its restriction to the weights each fused backward end is responsible for
    # (F.front.bwd: L0 of both nets + level 1's stride-2 convs; F.back.bwd: last expanding block + head) and to the rest
    def span(convs):
        m = torch.zeros_like(g)
        for c in convs:
            for v in (c.dkernel, c.dbias):
                off = (v.data_ptr() - pm.flat_grads.data_ptr()) // 4
                m[off:off + v.numel()] = 1
        return m
    q, o = pm.net['query'].layers, pm.net['obs'].layers
    front = span([q[0], o[0], q[1].convs()[0][0], o[1].convs()[0][0]])
    back = span([c for c, _ in q[-2].convs()] + [q[-1]])
    dirs = [g, g * front, g * back, g * (1 - front) * (1 - back)]
    assert all(float(d.norm()) > 0 for d in dirs)
    w0 = pm.flat_params.detach().clone()
    for d in [d / d.norm() for d in dirs]:
        eps = 0.6 * float(w0.double().norm()) / float(np.sqrt(w0.numel()))     # unit direction over 3.4 M weights: ~3e-4 rms(w) each
        vals = []
        for sgn in (1.0, -1.0):
            with torch.no_grad():
                pm.flat_params.copy_((w0.double() + sgn * eps * d).float())
            pm.mark_weights_updated()
            vals.append(value())
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float((g * d).sum())
        assert abs(fd - an) <= 0.05 * abs(an) + 1e-7, (loss, fd, an)
    with torch.no_grad():
        pm.flat_params.copy_(w0)
    pm.mark_weights_updated()


def test_train_step_without_observation_path():
    """use_obs = False (nlt/models/nlt.py:176-177) on the GPU: loss and every query-net gradient vs the oracle; the
    observation net's gradients stay zero."""
    om, pm = make_pair(depth=256, uv=64, im=32, loss='l2', seed=6, use_obs=False)
    pm.build('cuda')
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=51)
    po, go_, _, _ = om.call(batch, 'train', nn_list=nn)
    lo = om.compute_loss(po, go_, keep_batch=True).sum() / 2
    grads = torch.autograd.grad(lo, om.parameters(), allow_unused=True)
    pred, gt, kw, _ = pm(to_device_batch(batch, nn), mode='train')
    lp = pm.compute_loss(pred, gt, keep_batch=True).sum() / 2
    pm.flat_params.grad = None
    lp.backward()
    torch.cuda.synchronize()
    assert abs(float(lp.detach()) - float(lo.detach())) <= 1e-5 * abs(float(lo.detach()))
    it = iter(grads)
    n_q = sum(len(lw) for lw in om.wq)
    for i, c in enumerate(pm._conv_layers()):
        for name in ('dkernel', 'dbias'):
            g = next(it)
            got = getattr(c, name).cpu()
            if i < n_q:
                assert float((got - g).norm()) <= TENSOR_TOL * float(g.norm()), (i, name)
            else:
                assert g is None and not got.any(), (i, name)


def test_clipnorm_train_step_on_the_gpu():
    """mgm > 0: three steps with Keras clipnorm against the oracle (tf.clip_by_norm per variable before Adam)."""
    om, pm = make_pair(depth=256, uv=64, im=32, loss='l2', seed=8)
    pm.build('cuda')
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=52)
    db = to_device_batch(batch, nn)
    opt_o = O.KerasAdamAMSGrad(om.parameters(), 1e-3)
    opt_p = nlt_amd.optim.AdamAMSGrad(pm, 1e-3, clipnorm=1e-3)
    for step in range(3):
        lo, go = O.train_step(om, opt_o, batch, global_bs=2, nn_list=nn, clipnorm=1e-3)
        lp, _ = trainvali.distributed_train_step(pm, db, opt_p, global_bs=2)
        torch.cuda.synchronize()
        ref = flat_oracle_grads(pm, go)
        assert float((pm.flat_params.grad - ref).norm() / ref.norm()) < 5e-4, step
    worst = max(float((po.detach() - c.kernel.cpu()).abs().max()) for po, c in zip(om.parameters()[::2], pm._conv_layers()))
    assert worst < 2e-4, worst


@pytest.mark.parametrize('uv,k', [(256, 1), (256, 2)])
def test_backward_plan_is_bit_reproducible_given_the_texel_gradient(uv, k):
    """DESIGN section 3 / INTEGRATION: the only run-to-run variation of a train step comes from the float atomics of the resampler /
    resize adjoints and the Barron loss.  Given the SAME gradient w.r.t. the rendered texels the whole backward plan -- fused ends,
    backward-data launches, the LDS-tiled / row-walking / narrow weight gradients with their slice reductions, the weight-gradient
    side stream -- fills the flat bucket bit for bit the same, eager or replayed from its launch tape.  (UV >= 256 at depth 256: a level
    narrower than 4 texels -- depth 1024 at 256^2, toy sizes -- takes the first-generation weight-gradient kernel, whose partial
    sums meet in float atomics: one bias gradient then varies in its last bit, measured 5.8e-11 at UV 128.)"""
    _, pm = make_pair(depth=256, uv=uv, im=uv // 2, loss='l2', seed=31)
    pm.build('cuda')
    batch = to_device_batch(*O.synth_batch(2, uv, uv, uv // 2, uv // 2, uv // 2, uv // 2, k=k, seed=32))
    base, cvis, lvis, nn_base, nn_rgb = batch[1], batch[2], batch[3], batch[8], batch[9]
    g = torch.Generator(device='cuda').manual_seed(5)
    dpred = (torch.rand((2, uv, uv, 3), device='cuda', generator=g) - 0.5).contiguous()
    runs = []
    for i in range(5):                                          # first sights eager (+ plan-time trials), then recorded, then replays
        with torch.no_grad():
            pm._render(base, cvis, lvis, batch[4], nn_rgb, nn_base, None, None, False, inference=False)
            pm.flat_grads.zero_()
            pm.plan.backward(dpred, base, cvis, lvis, nn_rgb, nn_base, None, generation=pm.plan.generation)
        torch.cuda.synchronize()
        runs.append(pm.flat_grads.clone())
    assert float(runs[0].abs().max()) > 0
    assert pm.plan.tape_replays >= 1
    for r in runs[1:]:
        assert torch.equal(r, runs[0])
