"""-m gpu: the config branches the released .ini files leave off -- act = elu, norm = pixel / layer / batch, pool = max / avg,
upconv (nlt/networks/elements.py:42-56,69-94,103-121) -- on csrc/branches.hip, csrc/norms.hip + the layer-by-layer path (nlt_amd/generic.py):
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


@pytest.mark.parametrize('kind,name', [(C.NORM_LAYER, 'layer'), (C.NORM_BATCH, 'batch')])
@pytest.mark.parametrize('shape', [(2, 6, 5, 3), (1, 33, 17, 16), (3, 4, 4, 40), (1, 9, 7, 256), (1, 2, 3, 1024), (2, 96, 96, 32)])
def test_layer_and_batch_norm_forward_backward(kind, name, shape):
    """csrc/norms.hip vs torch autograd on the oracle's restatement (elements.py:51-56): y, dx, and dgamma / dbeta
    ACCUMULATED on top of what the gradient views already hold; two runs are bit-identical (fixed-order reductions)."""
    gen = torch.Generator().manual_seed(sum(shape) + kind)
    c = shape[-1]
    x = torch.randn(shape, generator=gen) * 2 + 0.5
    g = torch.randn(shape, generator=gen)
    gamma = torch.rand(c, generator=gen) + 0.5
    beta = torch.rand(c, generator=gen) - 0.5
    mean, var = torch.randn(c, generator=gen) * 0.1, torch.rand(c, generator=gen) + 0.5
    if kind == C.NORM_LAYER:
        f = lambda xx, gg, bb: O.layer_norm(xx, gg, bb)
    else:
        f = lambda xx, gg, bb: (xx - mean) * torch.rsqrt(var + O.NORM_EPS) * gg + bb
    xx, gg, bb = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    ref = f(xx, gg, bb)
    rdx, rdg, rdb = torch.autograd.grad(ref, (xx, gg, bb), g)
    d = lambda t: t.cuda()
    y = C.norm_forward(kind, d(x), d(gamma), d(beta), d(mean), d(var), O.NORM_EPS)
    outs = []
    for _ in range(2):
        dg, db = torch.full((c,), 2.0, device='cuda'), torch.full((c,), -1.0, device='cuda')
        dx = C.norm_backward(kind, d(g), d(x), d(gamma), d(mean), d(var), O.NORM_EPS, dg, db)
        torch.cuda.synchronize()
        outs.append((dx.cpu(), dg.cpu(), db.cpu()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    dx, dg, db = outs[0]
    assert rel_l2(y.cpu(), ref.detach()) <= 1e-6
    assert rel_l2(dx, rdx) <= 2e-5 and rel_l2(dg - 2.0, rdg) <= 2e-5 and rel_l2(db + 1.0, rdb) <= 2e-5
    if kind == C.NORM_BATCH:
        # the reference's run-time form: moving statistics at their initial (0, 1)
        y0 = C.norm_forward(kind, d(x), d(gamma), d(beta), torch.zeros(c, device='cuda'), torch.ones(c, device='cuda'), O.NORM_EPS)
        assert rel_l2(y0.cpu(), O.batch_norm_inference(x, gamma, beta)) <= 1e-6


def test_norm_refuses_more_than_1024_channels():
    x = torch.zeros(1, 2, 2, 1100, device='cuda')
    with pytest.raises(C.NLTError):
        C.norm_forward(C.NORM_LAYER, x, torch.ones(1100, device='cuda'), torch.zeros(1100, device='cuda'), None, None, 1e-3)


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


BRANCHES = [dict(act='elu'), dict(norm='pixel'), dict(pool='max'), dict(pool='avg', act='elu', norm='pixel'),
            dict(norm='layer'), dict(norm='batch'), dict(norm='layer', pool='avg', act='relu')]


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
