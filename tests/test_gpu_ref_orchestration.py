"""-m gpu: the HIP model against outputs of THE REFERENCE'S OWN MODEL CODE (tests/golden/ref_orchestration.npz, made by
tests/golden/make_ref_orchestration.py: nlt/models/nlt.py etc. imported from the reference tree and executed under the
test-side TensorFlow shim).  Modes train / vali / test, obs_override (the fused inference plan; feat_agg from
nlt_test.extract_feat), use_obs = False, skip_connect_base = False, depth 64 / 256 / 1024.  Bar: 1e-4 rel-L2."""
import numpy as np
import pytest
import torch

from gpu_util import make_pair, to_device_batch
from test_oracle_ref_orchestration import CASES, G, case, inputs, rel

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize('name', CASES)
def test_hip_model_equals_the_reference_s_own_model_code(name):
    from nlt_amd import nlt_test
    c = case(name)
    om, pm = make_pair(depth=c['depth'], uv=c['uv'], im=c['im'], seed=c['wseed'], use_obs=c['use_obs'],
                       skip_connect_base=c['skip_connect_base'])
    pm.build('cuda')
    batch, nn, train = inputs(c)
    kw = {}
    if c['override']:
        agg = nlt_test.extract_feat(pm, [to_device_batch(b, n_) for b, n_ in train])
        for l, a in enumerate(agg):
            fs = max(1, a.shape[1] // 16) if c['depth'] >= 1024 else 1
            assert rel(a[:, ::fs, ::fs].cpu(), G['%s/feat_agg_%d' % (name, l)]) <= TOL
        kw['obs_override'] = agg
    db = to_device_batch(batch, nn)
    for _ in range(3):                                  # plan-time trials, tape record, replay
        with torch.no_grad():
            pred_c, gt_c, _, vis = pm.call(db, c['mode'], **kw)
    torch.cuda.synchronize()
    if c['override']:
        assert pm.plan._ovr is not None                 # it ran on the fused query-only plan
    st = c['st']
    assert rel(vis['pred'][:, ::st, ::st].cpu(), G[name + '/pred']) <= TOL
    assert abs(float(vis['pred'].double().norm()) / float(G[name + '/pred_norm']) - 1) <= TOL
    assert rel(pred_c.cpu(), G[name + '/pred_camspc']) <= TOL
    assert rel(vis['base_camspc'].cpu(), G[name + '/base_camspc']) <= 1e-6
    if c['mode'] != 'test':
        assert rel(gt_c.cpu(), G[name + '/gt_camspc']) <= 1e-6
        per = pm.compute_loss(pred_c, gt_c, keep_batch=True)
        assert np.allclose(per.detach().cpu().numpy(), G[name + '/loss_per_example'], rtol=1e-4)
    else:
        assert gt_c is None
