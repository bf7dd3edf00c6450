"""-m gpu: the config branches the released .ini files leave off -- act = elu, norm = pixel, pool = max / avg, upconv
(nlt/networks/elements.py:42-48,69-94,103-121) -- on csrc/branches.hip + the layer-by-layer path (nlt_amd/generic.py):
each layer's forward / backward against torch-CPU autograd on the oracle's restatement, then whole models (forward and
one train step, every weight gradient) against oracle.OracleModel with the same branch."""
import numpy as np
import pytest
import torch

import nlt_amd
from nlt_amd import capi as C
from nlt_amd.models import get_model_class
from oracle import nlt_oracle as O
from oracle import tf_ops as T
from gpu_util import rel_l2, to_device_batch

pytestmark = pytest.mark.gpu


def _grad(fn, x, g):
    xx = x.clone().requires_grad_(True)
    return torch.autograd.grad(fn(xx), xx, g)[0]


@pytest.mark.parametrize('kind,alpha', [(C.ACT_LRELU, 0.3), (C.ACT_LRELU, 0.0), (C.ACT_ELU, 1.0)])
def test_activation_forward_backward(kind, alpha):
    x = torch.randn(3, 9, 7, 10)
    g = torch.randn_like(x)
    f = (lambda v: torch.nn.functional.elu(v, alpha)) if kind == C.ACT_ELU else (lambda v: T.leaky_relu(v, alpha))
    y = C.act_forward(x.cuda(), kind, alpha)
    dx = C.act_backward(g.cuda(), y, kind, alpha)
    assert float((y.cpu() - f(x)).abs().max()) <= 2e-7 and float((dx.cpu() - _grad(f, x, g)).abs().max()) <= 1e-6


@pytest.mark.parametrize('c', [3, 16, 40])
def test_pixelnorm_forward_backward(c):
    x = torch.randn(2, 6, 5, c)
    g = torch.randn_like(x)
    y = C.pixelnorm_forward(x.cuda())
    dx = C.pixelnorm_backward(g.cuda(), x.cuda())
    assert rel_l2(y.cpu(), O.pixel_norm(x)) <= 1e-6 and rel_l2(dx.cpu(), _grad(O.pixel_norm, x, g)) <= 1e-5


@pytest.mark.parametrize('kind,name', [(C.POOL_MAX, 'max'), (C.POOL_AVG, 'avg')])
def test_pool_forward_backward(kind, name):
    x = torch.randn(2, 8, 12, 5)
    g = torch.randn(2, 4, 6, 5)
    y = C.pool2x2_forward(x.cuda(), kind)
    dx = C.pool2x2_backward(g.cuda(), x.cuda(), kind)
    assert torch.equal(y.cpu(), O.pool2x2(x, name).contiguous()) or rel_l2(y.cpu(), O.pool2x2(x, name)) <= 1e-7
    assert rel_l2(dx.cpu(), _grad(lambda v: O.pool2x2(v, name), x, g)) <= 1e-7
    with pytest.raises(C.NLTError):
        C.pool2x2_forward(torch.zeros(1, 7, 8, 4, device='cuda'), kind)              # odd size: TF 'same' would pad


def _pair(**kw):
    om = O.OracleModel(depth=32, uvh=128, uvw=128, imh=64, imw=64, seed=2, **kw)
    pm = get_model_class('nlt')(nlt_amd.make_config(depth=32, uvh=128, uvw=128, imh=64, imw=64, **kw))
    pm.load_weights(om.numpy_weights())
    pm.register_trainable()
    return om, pm


BRANCHES = [dict(act='elu'), dict(norm='pixel'), dict(pool='max'), dict(pool='avg', act='elu', norm='pixel')]


@pytest.mark.parametrize('kw', BRANCHES, ids=lambda kw: '+'.join('%s=%s' % x for x in kw.items()))
def test_branch_model_forward_and_train_step_vs_oracle(kw):
    om, pm = _pair(loss='l2', **kw)
    assert pm.generic
    pm.build('cuda')
    batch, nn = O.synth_batch(2, 128, 128, 64, 64, 64, 64, k=2, seed=9)
    db = to_device_batch(batch, nn)
    with torch.no_grad():
        ref = om.call(batch, 'vali', nn_list=nn)
    got = pm.call(db, 'vali', want_indices=True)
    torch.cuda.synchronize()
    assert rel_l2(got[3]['pred'].cpu(), ref[3]['pred']) <= 1e-4 and rel_l2(got[0].cpu(), ref[0]) <= 1e-4
    po, go, _, _ = om.call(batch, 'train', nn_list=nn)
    lo = om.compute_loss(po, go, keep_batch=True).sum() / 2
    grads = torch.autograd.grad(lo, om.parameters())
    pred, gt, _, _ = pm(db, mode='train')
    lp = pm.compute_loss(pred, gt, keep_batch=True).sum() / 2
    pm.flat_params.grad = None
    lp.backward()
    torch.cuda.synchronize()
    assert abs(float(lp.detach()) - float(lo.detach())) <= 1e-5 * abs(float(lo.detach()))
    it = iter(grads)
    worst = 0.0
    for c in pm._conv_layers():
        for name in ('dkernel', 'dbias'):
            g = next(it)
            worst = max(worst, float((getattr(c, name).cpu() - g).norm() / (g.norm() + 1e-30)))
    assert worst <= 5e-3, worst                                   # (max-pool / activation kinks: see test_gpu_baseline_sizes.py)
    # and a full optimizer step runs on this path
    opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
    loss, _ = nlt_amd.trainvali.distributed_train_step(pm, db, opt, 2)
    assert np.isfinite(float(loss))
