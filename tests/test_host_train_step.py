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
