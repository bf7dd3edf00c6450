"""CPU: the oracle's orchestration -- and the product's host orchestration on emulated kernels -- against outputs of THE
REFERENCE'S OWN MODEL CODE (nlt/models/{base,nlt}.py, nlt/networks/*, nlt/util/img.py, nlt/losses.py:L2,
nlt/nlt_test.py:extract_feat), imported from /root/reference and executed by tests/golden/make_ref_orchestration.py under the
test-side TensorFlow shim (tests/tf_shim/: each TF primitive delegates to oracle/tf_ops.py).

This is what makes `oracle/nlt_oracle.py:OracleModel` (call / _call / compute_loss; modes; obs_override; use_obs = False;
skip_connect_base = False; depth 64 / 256 / 1024 incl. the bottleneck self-concat) and the feature aggregation
"orchestration: pinned (reference code executed)".  The TF kernels behind the primitives stay unpinned.
The fixture travels; /root/reference is not read here."""
import os

import numpy as np
import pytest
import torch

import nlt_amd
from nlt_amd.models import get_model_class
from oracle import nlt_oracle as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_orchestration.npz'))
CASES = sorted({k.split('/')[0] for k in G.files if k.startswith('d')})


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def case(name):
    depth, uv, im, cam, n, wseed, bseed, st, use_obs, skip_base, override = (int(x) for x in G[name + '/meta'])
    return dict(depth=depth, uv=uv, im=im, cam=cam, n=n, wseed=wseed, bseed=bseed, st=st, use_obs=bool(use_obs),
                skip_connect_base=bool(skip_base), override=bool(override), mode=str(G[name + '/mode']))


def inputs(c):
    batch, nn = O.synth_batch(c['n'], c['uv'], c['uv'], c['cam'], c['cam'], c['im'], c['im'], k=1, seed=c['bseed'])
    train = None
    if c['override']:
        train = [O.synth_batch(nf, c['uv'], c['uv'], c['cam'], c['cam'], c['im'], c['im'], k=1, seed=c['bseed'] + 1000 + nf)
                 for nf in ((2, 3) if c['depth'] < 1024 else (1, 2))]
    return batch, nn, train


def oracle_feat_agg(om, train):
    with torch.no_grad():
        feats = [om._call(torch.cat((b[1], b[2], b[3]), 3), [b[5] - b[1]], return_feats=True)[1] for b, _ in train]
    return [torch.cat([f[l] for f in feats], 0).mean(0, keepdim=True) for l in range(len(feats[0]))]


def check_outputs(name, c, pred, pred_c, base_c, gt_c, tol):
    st = c['st']
    assert rel(pred[:, ::st, ::st], G[name + '/pred']) <= tol
    assert abs(float(np.linalg.norm(np.asarray(pred, np.float64))) / float(G[name + '/pred_norm']) - 1) <= tol
    assert rel(pred_c, G[name + '/pred_camspc']) <= tol
    assert rel(base_c, G[name + '/base_camspc']) <= tol
    if c['mode'] != 'test':
        assert rel(gt_c, G[name + '/gt_camspc']) <= tol
    else:
        assert gt_c is None


@pytest.mark.parametrize('name', CASES)
def test_oracle_model_equals_the_reference_s_own_model_code(name):
    c = case(name)
    om = O.OracleModel(depth=c['depth'], uvh=c['uv'], uvw=c['uv'], imh=c['im'], imw=c['im'], seed=c['wseed'],
                       use_obs=c['use_obs'], skip_connect_base=c['skip_connect_base'])
    # the layer list the reference built (convnet.py:30-90) is the oracle's
    assert list(G[name + '/is_contracting']) == [int(x) for x in om.is_contracting]
    assert int(G[name + '/n_obs_layers']) == len(om.wo) == sum(om.is_contracting)
    batch, nn, train = inputs(c)
    override = None
    if c['override']:
        agg = oracle_feat_agg(om, train)
        for l, a in enumerate(agg):
            fs = max(1, a.shape[1] // 16) if c['depth'] >= 1024 else 1
            assert rel(a[:, ::fs, ::fs], G['%s/feat_agg_%d' % (name, l)]) <= 1e-6
            assert abs(float(a.double().norm()) / float(G['%s/feat_agg_norm_%d' % (name, l)]) - 1) <= 1e-6
        first = oracle_feat_agg(om, train[:1])                   # --n_obs_batches 1
        assert np.allclose([float(f.double().norm()) for f in first], G[name + '/feat_first_batch_only_norms'], rtol=1e-5)
        override = [a.expand(c['n'], -1, -1, -1) for a in agg]
    with torch.no_grad():
        pred_c, gt_c, kw, vis = om.call(batch, c['mode'], obs_override=override, nn_list=nn)
    assert (kw == {}) if c['mode'] != 'test' else kw is None
    check_outputs(name, c, vis['pred'].numpy(), pred_c.numpy(), vis['base_camspc'].numpy(), None if gt_c is None else gt_c.numpy(), 1e-6)
    if c['mode'] != 'test':
        with torch.no_grad():
            per = om.compute_loss(pred_c, gt_c, keep_batch=True)
            sc = om.compute_loss(pred_c, gt_c, keep_batch=False)
        assert np.allclose(per.numpy(), G[name + '/loss_per_example'], rtol=1e-5)
        assert np.allclose(float(sc), float(G[name + '/loss_scalar']), rtol=1e-5)


def test_bad_modes_raise_like_the_reference():
    assert 'bad_mode_raises_ValueError/training' in G.files and 'bad_mode_raises_ValueError/bogus' in G.files
    om = O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32)
    pm = get_model_class('nlt')(nlt_amd.make_config(depth=256, uvh=64, uvw=64, imh=32, imw=32))
    for m in (om, pm):
        with pytest.raises(ValueError):
            m.call((None,) * 11, 'training')


@pytest.mark.parametrize('name', [n for n in CASES if not n.startswith('d1024')] + ['d1024_override'])
def test_product_host_orchestration_equals_the_reference_s_own_model_code(monkeypatch, name):
    """The product's plans (fused ends, fused obs_override, layer-by-layer) driven through the CPU emulation of the C ABI
    (tests/fake_capi.py) against the same fixture: the host side's buffer / concat / slice bookkeeping is the reference's."""
    import fake_capi
    from test_host_orchestration import make, cpu_batch
    fake_capi.install(monkeypatch)
    c = case(name)
    om, pm = make(c['depth'], c['uv'], c['im'], use_obs=c['use_obs'], skip_connect_base=c['skip_connect_base'])
    # `make` seeds the oracle with 1: take this fixture's weights instead
    om = O.OracleModel(depth=c['depth'], uvh=c['uv'], uvw=c['uv'], imh=c['im'], imw=c['im'], seed=c['wseed'],
                       use_obs=c['use_obs'], skip_connect_base=c['skip_connect_base'])
    for net in ('query', 'obs'):
        for layer, lw in zip(pm.net[net].layers, om.numpy_weights()[net]):
            convs = [layer] if hasattr(layer, 'set_weights') else [cv for cv, _ in layer.convs()]
            for cv, (k, b) in zip(convs, lw):
                cv.kernel, cv.bias = torch.tensor(k), torch.tensor(b)
    batch, nn, train = inputs(c)
    kw = {}
    if c['override']:
        from nlt_amd import nlt_test
        agg = nlt_test.extract_feat(pm, [cpu_batch(b, n_) for b, n_ in train])
        for l, a in enumerate(agg):
            fs = max(1, a.shape[1] // 16) if c['depth'] >= 1024 else 1
            assert rel(a[:, ::fs, ::fs], G['%s/feat_agg_%d' % (name, l)]) <= 1e-5
        kw['obs_override'] = agg
    pred_c, gt_c, _, vis = pm.call(cpu_batch(batch, nn), c['mode'], **kw)
    check_outputs(name, c, vis['pred'].numpy(), pred_c.numpy(), vis['base_camspc'].numpy(), None if gt_c is None else gt_c.numpy(), 1e-5)
