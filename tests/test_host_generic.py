"""CPU: the layer-by-layer path (nlt_amd/generic.py) for the config branches the fused plan does not execute -- act = elu,
norm = pixel / layer / batch, pool = max / avg (+ upconv) -- driven through the TEST-ONLY C-ABI emulation against the oracle: forward,
and one train step's loss and every weight gradient (the hand-rolled backward tape: conv adjoints, concat splits, the
observation mean, the skip stack's fan-in).  Also: the generic path on the DEFAULT config equals the fused plan."""
import numpy as np
import pytest
import torch

import nlt_amd
from nlt_amd.models import get_model_class
from nlt_amd.models.nlt import _convs_of
from oracle import nlt_oracle as O
import fake_capi
from test_host_orchestration import cpu_batch, rel_l2


def make(depth, uv, im, **kw):
    om = O.OracleModel(depth=depth, uvh=uv, uvw=uv, imh=im, imw=im, seed=1, **kw)
    pm = get_model_class('nlt')(nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=im, imw=im, **kw))
    for name in ('query', 'obs'):
        for layer, lw in zip(pm.net[name].layers, om.numpy_weights()[name]):
            convs = _convs_of(layer)
            assert len(convs) == len(lw)
            for c, (k, b) in zip(convs, lw):
                c.kernel, c.bias = torch.tensor(k), torch.tensor(b)
                if hasattr(c, 'transpose'):
                    c.cin = k.shape[3] if c.transpose else k.shape[2]
                else:                                            # ChannelNorm: (gamma, beta)
                    c.c = k.shape[0]
                c.built = True
    return om, pm


BRANCHES = [dict(act='elu'), dict(norm='pixel'), dict(pool='max'), dict(pool='avg'), dict(act='elu', norm='pixel', pool='max'),
            dict(norm='layer'), dict(norm='batch'), dict(norm='layer', pool='avg', act='relu')]


@pytest.mark.parametrize('kw', BRANCHES, ids=lambda kw: '+'.join('%s=%s' % x for x in kw.items()))
def test_branch_forward_and_train_step_match_oracle(monkeypatch, kw):
    fake_capi.install(monkeypatch)
    om, pm = make(32, 64, 32, loss='l2', **kw)
    assert pm.generic
    pm.build('cpu'); pm.register_trainable()
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=3)
    cb = cpu_batch(batch, nn)
    with torch.no_grad():
        ref = om.call(batch, 'vali', nn_list=nn)
        got = pm.call(cb, 'vali')
    assert rel_l2(got[3]['pred'], ref[3]['pred']) < 1e-5 and rel_l2(got[0], ref[0]) < 1e-5 and rel_l2(got[1], ref[1]) < 1e-6
    po, go, _, _ = om.call(batch, 'train', nn_list=nn)
    lo = om.compute_loss(po, go, keep_batch=True).sum() / 2
    grads = torch.autograd.grad(lo, om.parameters())
    pred, gt, kw2, _ = pm(cb, mode='train')
    lp = pm.compute_loss(pred, gt, keep_batch=True).sum() / 2
    pm.flat_params.grad = None
    lp.backward()
    assert abs(float(lp.detach()) - float(lo.detach())) <= 1e-5 * abs(float(lo.detach()))
    it = iter(grads)
    for i, c in enumerate(pm._conv_layers()):
        for name in ('dkernel', 'dbias'):
            g = next(it)
            assert float((getattr(c, name) - g).norm()) <= 3e-4 * float(g.norm()) + 1e-9, (i, name)


def test_unsupported_norms_say_why():
    with pytest.raises(NotImplementedError, match='tf.contrib'):
        get_model_class('nlt')(nlt_amd.make_config(depth=32, norm='instance'))
    with pytest.raises(NotImplementedError):
        get_model_class('nlt')(nlt_amd.make_config(depth=32, act='gelu'))


def test_generic_path_on_the_default_config_equals_the_fused_plan(monkeypatch):
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 32, loss='l2')
    assert not pm.generic
    pm.build('cpu'); pm.register_trainable()
    batch, nn = O.synth_batch(1, 64, 64, 32, 32, 32, 32, k=2, seed=5)
    cb = cpu_batch(batch, nn)
    a = pm.call(cb, 'test')
    pm.generic = True
    b = pm.call(cb, 'test')
    assert rel_l2(b[3]['pred'], a[3]['pred']) < 1e-5 and rel_l2(b[0], a[0]) < 1e-5
