"""CPU: the product's train step (autograd glue -> hand-ordered backward plan -> flat gradient
bucket -> fused Keras Adam-AMSGrad) driven through the TEST-ONLY C-ABI emulation, against the
oracle's torch-autograd train step.  Guards the backward ORCHESTRATION (which buffer holds which
gradient, adjoint modes, weight slices, accumulation order, mask placement); kernel arithmetic is
checked by the -m gpu tests."""
import numpy as np
import pytest
import torch

import nlt_amd
from nlt_amd import trainvali
from nlt_amd.models import get_model_class
from oracle import nlt_oracle as O
import fake_capi
from test_host_orchestration import make, cpu_batch


def flat_oracle_grads(om, pm, grads):
    """oracle grads (list in oracle.parameters() order) -> product flat-bucket layout."""
    out = torch.zeros_like(pm.flat_grads)
    it = iter(grads)
    for c in pm._conv_layers():
        for name in ('dkernel', 'dbias'):
            g = next(it)
            view = getattr(c, name)
            off = view.data_ptr() - pm.flat_grads.data_ptr()
            out.view(-1)[off // 4: off // 4 + g.numel()] = g.reshape(-1)
    return out


@pytest.mark.parametrize('loss,k,uv,cam,im', [('l2', 1, 64, 32, 32), ('l2', 3, 64, 16, 32), ('barron', 2, 64, 32, 32),
                                              ('barron,2e+0l2', 1, 64, 32, 32)])
def test_train_step_matches_oracle(monkeypatch, loss, k, uv, cam, im):
    fake_capi.install(monkeypatch)
    om, pm = make(256, uv, im, loss=loss)
    pm.build('cpu')
    pm.register_trainable()
    assert pm.n_params == 3368071 and pm.flat_params.numel() >= pm.n_params
    batch, nn = O.synth_batch(2, uv, uv, cam, cam, im, im, k=k, seed=7)
    opt_o = O.KerasAdamAMSGrad(om.parameters(), 1e-3)
    opt_p = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
    for step in range(2):
        lo, go = O.train_step(om, opt_o, batch, global_bs=4, nn_list=nn)
        lp, _ = trainvali.distributed_train_step(pm, cpu_batch(batch, nn), opt_p, global_bs=4)
        assert abs(float(lp) - float(lo)) <= 1e-5 * max(1.0, abs(float(lo)))
        ref = flat_oracle_grads(om, pm, go)
        got = pm.flat_params.grad
        rel = float((got - ref).norm() / ref.norm())
        assert rel < 2e-4, (step, rel)
        # weights after the fused Adam step track the oracle's
        for po, c in zip(om.parameters()[::2], pm._conv_layers()):
            assert float((po.detach() - c.kernel).abs().max()) < 2e-5


def test_train_step_with_the_winograd_launches_of_forward_and_backward_data(monkeypatch):
    """Every eligible stride-1 k2 launch -- forward convs and their backward-data adjoints (weight slices read in place from the
    layer's own array, mask / accumulate epilogue) -- given to csrc/conv_wino.hip's entry points: same loss and gradient."""
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 32, loss='l2')
    pm.build('cpu'); pm.register_trainable()
    pm.plan.autotune = False
    pm.plan._trial_wino = 32
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=12)
    opt_o = O.KerasAdamAMSGrad(om.parameters(), 1e-3)
    opt_p = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
    lo, go = O.train_step(om, opt_o, batch, global_bs=2, nn_list=nn)
    lp, _ = trainvali.distributed_train_step(pm, cpu_batch(batch, nn), opt_p, global_bs=2)
    assert abs(float(lp) - float(lo)) <= 1e-5 * max(1.0, abs(float(lo)))
    ref = flat_oracle_grads(om, pm, go)
    assert float((pm.flat_params.grad - ref).norm() / ref.norm()) < 2e-4
    ran = pm.plan._ran_wino
    assert {'L3.q.s1', 'L3.o.s1', 'L7.q.s1', 'bwd.L3.q.s1.dgrad', 'bwd.L3.o.s1.dgrad', 'bwd.L7.q.s1.dgrad'} <= ran, sorted(ran)
    assert not any('.s2' in l for l in ran)


def test_vali_step_and_no_grad_paths(monkeypatch):
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 32, loss='l2')
    pm.build('cpu'); pm.register_trainable()
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=8)
    with torch.no_grad():
        pred, gt, _, _ = om.call(batch, 'vali', nn_list=nn)
        ref = om.compute_loss(pred, gt, keep_batch=True).sum() / 2
    loss, vis = trainvali.distributed_vali_step(pm, cpu_batch(batch, nn), 2)
    assert abs(float(loss) - float(ref)) < 1e-5
    assert not vis['pred_camspc'].requires_grad


def test_tape_free_train_forward_backward_equals_the_autograd_path(monkeypatch):
    """Model.train_forward_backward (what trainvali.GraphedTrainStep captures in a hipGraph) fills the same flat
    gradient bucket and returns the same loss as call('train') + compute_loss + backward()."""
    fake_capi.install(monkeypatch)
    _, pm = make(256, 64, 32, loss='barron,2e+0l2')
    pm.build('cpu'); pm.register_trainable()
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=11)
    b = cpu_batch(batch, nn)
    pred, gt, kw, _ = pm(b, mode='train')
    loss = pm.compute_loss(pred, gt, keep_batch=True).sum() / 4
    pm.flat_params.grad = None
    loss.backward()
    ref = pm.flat_params.grad.clone()
    loss2, vis = pm.train_forward_backward(b, 4)
    loss = loss.detach()
    assert abs(float(loss2) - float(loss)) <= 1e-6 * abs(float(loss))
    assert torch.equal(pm.flat_grads, ref)
    assert set(vis) >= {'pred', 'pred_camspc', 'gt_camspc', 'base_camspc'}
    # the graphed-step object degrades to the eager step off the GPU
    step = trainvali.GraphedTrainStep(pm, nlt_amd.optim.AdamAMSGrad(pm, 1e-3), 4, warmup=0)
    l3, _ = step(b)
    assert step.graph is None and abs(float(l3) - float(loss)) <= 1e-6 * abs(float(loss))


def test_checkpoint_save_restore_resumes_training_bit_for_bit(monkeypatch, tmp_path):
    """trainvali.save_checkpoint / restore_checkpoint (nlt/trainvali.py:134-141,197): weights + Adam-AMSGrad slots +
    iteration count + global step.  Train 2 steps, save, train 2 more; a FRESH model + optimizer restored from the file and
    trained 2 steps must land on exactly the same weights; inference can restore the net alone; get_weights round-trips
    through load_weights; a different architecture is refused."""
    fake_capi.install(monkeypatch)
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=21)
    b = cpu_batch(batch, nn)

    def fresh(seed):
        _, pm = make(256, 64, 32, loss='l2')
        pm.build('cpu'); pm.register_trainable()
        with torch.no_grad():
            pm.flat_params.mul_(1.0 + 0.01 * seed)               # different initial weights per instance
        pm.mark_weights_updated()
        return pm, nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
    pm, opt = fresh(1)
    for _ in range(2):
        trainvali.distributed_train_step(pm, b, opt, 2)
    path = trainvali.save_checkpoint(str(tmp_path / 'ckpt-2.pt'), pm, opt, step=2)
    for _ in range(2):
        trainvali.distributed_train_step(pm, b, opt, 2)
    want = pm.flat_params.detach().clone()

    pm2, opt2 = fresh(2)                                        # other initial weights: everything must come from the file
    assert trainvali.restore_checkpoint(path, pm2, opt2) == 2 and opt2.t == 2
    for _ in range(2):
        trainvali.distributed_train_step(pm2, b, opt2, 2)
    assert torch.equal(pm2.flat_params.detach(), want)

    pm3, _ = fresh(3)
    assert trainvali.restore_checkpoint(path, pm3) == 2         # net alone (nlt_test.py:92-97)
    pm4, _ = fresh(4)
    pm4.load_weights(pm3.get_weights())
    assert torch.equal(pm4.flat_params.detach(), pm3.flat_params.detach())
    w = pm3.get_weights()
    w['query'][1][0] = (w['query'][1][0][0][..., :8], w['query'][1][0][1][:8])
    with pytest.raises(ValueError, match='expects kernel'):
        pm4.load_weights(w)
    _, deep = make(1024, 64, 32, loss='l2')
    deep.build('cpu')
    with pytest.raises(ValueError, match='different architecture'):
        trainvali.restore_checkpoint(path, deep)
    # optimizer slots of another size are refused before anything is overwritten; a resident batch trains like its float twin
    sd = opt.state_dict()
    sd['v'] = sd['v'][:10]
    with pytest.raises(ValueError, match="optimizer state 'v'"):
        opt2.load_state_dict(sd)
    # advisor r05 (high): the checkpoint must not depend on the flat bucket's slot ORDER.  A model built with the other
    # gradient-range layout (NLT_GRAD_RANGES=2: r01-r04's order) restores the same file to the same per-layer weights and the
    # same optimizer slots, and continues to the same weights.
    monkeypatch.setenv('NLT_GRAD_RANGES', '2')
    pm5, opt5 = fresh(5)
    monkeypatch.delenv('NLT_GRAD_RANGES')
    assert pm5.bucket_ranges != pm2.bucket_ranges and [s_[0] for s_ in pm5._slots] != [s_[0] for s_ in pm2._slots]
    assert trainvali.restore_checkpoint(path, pm5, opt5) == 2
    for _ in range(2):
        trainvali.distributed_train_step(pm5, b, opt5, 2)
    for a_, b_ in zip(pm5.bucket_to_variables(pm5.flat_params), pm2.bucket_to_variables(pm2.flat_params)):
        assert torch.equal(a_, b_)
    for k_ in ('m', 'v', 'vhat'):
        for a_, b_ in zip(opt5.state_dict()[k_], opt2.state_dict()[k_]):
            assert torch.equal(a_, b_)
    # a format-1 file (raw flat bucket) is refused unless the caller vouches for its layout; vouched for, it converts
    raw = lambda o_: {'t': o_.t, 'm': o_.m.clone(), 'v': o_.v.clone(), 'vhat': o_.vhat.clone()}
    old = str(tmp_path / 'old.pt')
    torch.save({'format': 'nlt_amd-ckpt-1', 'step': 7, 'optimizer': raw(opt2),
                'net': {'flat_params': pm2.flat_params.detach().clone(),
                        'slots': [(tuple(c.kernel.shape), tuple(c.bias.shape)) for c in pm2._conv_layers()]}}, old)
    pm6, opt6 = fresh(6)
    with pytest.raises(ValueError, match='format-1'):
        trainvali.restore_checkpoint(old, pm6, opt6)
    assert trainvali.restore_checkpoint(old, pm6, opt6, legacy_layout=True) == 7
    assert torch.equal(pm6.flat_params.detach(), pm2.flat_params.detach()) and torch.equal(opt6.vhat, opt2.vhat)
    # the file is loaded with weights_only=True: a pickle that would run code is refused
    import pickle

    class Evil:
        def __reduce__(self):
            return (print, ('code ran',))
    bad = str(tmp_path / 'evil.pt')
    torch.save({'format': 'nlt_amd-ckpt-1', 'step': 1, 'net': Evil(), 'optimizer': None}, bad)
    with pytest.raises(pickle.UnpicklingError):
        trainvali.restore_checkpoint(bad, pm3)


def test_backward_after_a_later_forward_raises(monkeypatch):
    """The plan keeps ONE set of activations: a backward whose forward has been overwritten (second micro-batch, a vali
    call in between) must not silently pair its inputs with the other pass's activations."""
    fake_capi.install(monkeypatch)
    _, pm = make(256, 64, 32, loss='l2')
    pm.build('cpu'); pm.register_trainable()
    b1 = cpu_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=31))
    b2 = cpu_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=32))
    p1, g1, _, _ = pm(b1, mode='train')
    l1 = pm.compute_loss(p1, g1, keep_batch=True).sum()
    pm(b2, mode='vali')                                          # overwrites the activations l1's backward needs
    with pytest.raises(RuntimeError, match='overwritten by a later forward'):
        l1.backward()
    p1, g1, _, _ = pm(b1, mode='train')                         # the normal order still works
    pm.compute_loss(p1, g1, keep_batch=True).sum().backward()


def test_adapters_refuse_mismatched_shapes(monkeypatch):
    """Raw kernels take element counts: the adapters compare shapes first (rgb_camspc vs fg_camspc)."""
    from nlt_amd import capi as C
    a, b = torch.zeros(2, 8, 8, 3), torch.zeros(2, 4, 4, 3)
    for fn, args in ((C.mul_forward, (a, b)), (C.l2_loss_forward, (a, b)), (C.l2_loss_backward, (a, b, torch.zeros(2))),
                     (C.barron_loss, (a, b, False)), (C.scale_rows, (a, torch.zeros(3)))):
        with pytest.raises(C.NLTError):
            fn(*args)


def test_clipnorm_train_step_matches_oracle(monkeypatch):
    """mgm > 0 (nlt/trainvali.py:122-127): Keras clipnorm = tf.clip_by_norm per variable, before the Adam update."""
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 32, loss='l2')
    pm.build('cpu'); pm.register_trainable()
    pm.config.set('DEFAULT', 'mgm', '1e-3')
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=41)
    opt_o = O.KerasAdamAMSGrad(om.parameters(), 1e-3)
    # default = what TF 2.2's tape.gradient + apply_gradients loop does with clipnorm: nothing (a warning says so)
    with pytest.warns(UserWarning, match='does not apply clipnorm'):
        assert trainvali.make_optimizer(pm, pm.config).clipnorm is None
    pm.config.set('DEFAULT', 'mgm_apply', 'true')               # opt-in: per-variable tf.clip_by_norm
    opt_p = trainvali.make_optimizer(pm, pm.config)
    assert opt_p.clipnorm == 1e-3
    clipped = 0
    for step in range(2):
        lo, go = O.train_step(om, opt_o, batch, global_bs=2, nn_list=nn, clipnorm=1e-3)
        lp, _ = trainvali.distributed_train_step(pm, cpu_batch(batch, nn), opt_p, global_bs=2)
        ref = flat_oracle_grads(om, pm, go)
        assert float((pm.flat_params.grad - ref).norm() / ref.norm()) < 2e-4
        clipped += sum(float(g.norm()) > 0.99e-3 for g in go)
        for po, c in zip(om.parameters()[::2], pm._conv_layers()):
            assert float((po.detach() - c.kernel).abs().max()) < 2e-5
    assert clipped > 0                                           # the clip was active on some tensors


def test_train_step_without_observation_path(monkeypatch):
    """use_obs = False (nlt/models/nlt.py:176-177): the query net alone carries the gradient; the observation net's
    gradients stay zero (TF: `None`, skipped by apply_gradients)."""
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 32, loss='l2', use_obs=False)
    pm.build('cpu'); pm.register_trainable()
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=51)
    po, go_, _, _ = om.call(batch, 'train', nn_list=nn)
    lo = om.compute_loss(po, go_, keep_batch=True).sum() / 2
    grads = torch.autograd.grad(lo, om.parameters(), allow_unused=True)
    pred, gt, kw, _ = pm(cpu_batch(batch, nn), mode='train')
    lp = pm.compute_loss(pred, gt, keep_batch=True).sum() / 2
    pm.flat_params.grad = None
    lp.backward()
    assert abs(float(lp.detach()) - float(lo.detach())) <= 1e-5 * abs(float(lo.detach()))
    it = iter(grads)
    n_q = sum(len(lw) for lw in om.wq)
    for i, c in enumerate(pm._conv_layers()):
        for name in ('dkernel', 'dbias'):
            g = next(it)
            got = getattr(c, name)
            if i < n_q:
                assert g is not None and float((got - g).norm()) <= 2e-4 * float(g.norm()), (i, name)
            else:
                assert g is None and not got.any(), (i, name)


def test_mask_conditioned_oracle_is_the_plain_oracle_on_its_own_branches(monkeypatch):
    """The instrument behind the per-tensor gradient bars of tests/test_gpu_baseline_sizes.py: `OracleModel.act_masks`
    dictates the branch of every LeakyReLU (keys / shapes as `gpu_util.hip_activation_masks` reads them off the plan's
    buffers).  With the branches the forward took anyway the float64 oracle must not change at all; with ONE texel of one
    deep activation flipped, the gradient of the conv feeding it moves -- that is the discontinuity the conditioning removes."""
    from gpu_util import hip_activation_masks
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 32, loss='l2')
    pm.build('cpu'); pm.register_trainable()
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=7)
    lp, _ = pm.train_forward_backward(cpu_batch(batch, nn), 2)
    masks = hip_activation_masks(pm)
    assert set(masks) == ({('q', l) for l in range(1, 13)} | {('o', l, j) for l in range(1, 7) for j in range(2)})
    om64 = O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32, loss='l2', dtype=torch.float64)
    for a, b in zip(om64.parameters(), om.parameters()):
        a.data.copy_(b.data.double())

    def grads(m):
        om64.act_masks = m
        b = tuple(t.double() if torch.is_tensor(t) else t for t in batch)
        po, go, _, _ = om64.call(b, 'train', nn_list=[(a.double(), c.double()) for a, c in nn])
        lo = om64.compute_loss(po, go, keep_batch=True).sum() / 2
        return lo.detach(), torch.autograd.grad(lo, om64.parameters())
    l0, g0 = grads(None)
    l1, g1 = grads(masks)
    assert float(l0) == float(l1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
    assert abs(float(lp) - float(l0)) <= 1e-6 * abs(float(l0))
    it = iter(g1)
    for c in pm._conv_layers():
        for nm in ('dkernel', 'dbias'):
            g = next(it)
            assert float((getattr(c, nm).double() - g).norm()) <= 1e-5 * float(g.norm())
    flipped = dict(masks)
    m0, m1 = masks[('q', 5)]
    m1 = m1.clone(); m1[0, 0, 0, 0] = ~m1[0, 0, 0, 0]
    flipped[('q', 5)] = (m0, m1)
    _, g2 = grads(flipped)
    assert not all(torch.equal(a, b) for a, b in zip(g1, g2))
