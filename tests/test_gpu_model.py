"""-m gpu: full Model.call forward (fused RenderPlan on HIP kernels) vs the CPU oracle.
Tolerance (BASELINE.json): rel-L2 <= 1e-4 on the rendered texels; integer UV indices bit-exact."""
import os
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from oracle import nlt_oracle as O
from oracle import tf_ops as T
from gpu_util import rel_l2, make_pair, to_device_batch

pytestmark = pytest.mark.gpu
TOL = 1e-4
G = os.path.join(os.path.dirname(__file__), 'golden')


def _compare(om, pm, batch, nn, algo=C.ALGO_AUTO, mode='train'):
    with torch.no_grad():
        o_pred_c, o_gt_c, _, o_vis = om.call(batch, mode, nn_list=nn)
    pm.conv_algo = algo
    p_pred_c, p_gt_c, _, p_vis = pm.call(to_device_batch(batch, nn), mode, want_indices=True)
    torch.cuda.synchronize()
    assert rel_l2(p_vis['pred'].cpu(), o_vis['pred']) <= TOL
    assert rel_l2(p_pred_c.cpu(), o_pred_c) <= TOL
    assert rel_l2(p_vis['base_camspc'].cpu(), o_vis['base_camspc']) <= 1e-6
    if mode != 'test':
        assert rel_l2(p_gt_c.cpu(), o_gt_c) <= 1e-6
    fx, fy, inside = T.resampler_indices(o_vis['warp_px'].numpy(), om.uvh, om.uvw)
    idx = p_vis['uv_indices'].cpu().numpy()
    np.testing.assert_array_equal(idx[..., 0], fx)
    np.testing.assert_array_equal(idx[..., 1], fy)
    np.testing.assert_array_equal(idx[..., 2], inside.astype(np.int32))
    return p_vis


@pytest.mark.parametrize('k', [1, 2, 4])
@pytest.mark.parametrize('algo', [C.ALGO_AUTO, C.ALGO_DIRECT])
def test_forward_depth256_64(k, algo):
    om, pm = make_pair(depth=256, uv=64, im=32, seed=k)
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=k, seed=10 + k)
    _compare(om, pm, batch, nn, algo)


def test_forward_relight_only_identity_warp_128():
    # BASELINE config 2 shape family (relight only: identity warp), reduced to 128^2 for the CPU oracle
    om, pm = make_pair(depth=256, uv=128, im=128, seed=3)
    batch, nn = O.synth_batch(2, 128, 128, 128, 128, 128, 128, k=1, seed=4, identity_warp=True)
    vis = _compare(om, pm, batch, nn)
    # identity warp: camera-space prediction is the UV prediction itself
    np.testing.assert_array_equal(vis['pred_camspc'].cpu().numpy(), vis['pred'].cpu().numpy())


def test_forward_resized_camera_space_and_test_mode():
    om, pm = make_pair(depth=256, uv=64, im=48, seed=5)      # warp res 32 != im res 48 -> resize path
    batch, nn = O.synth_batch(1, 64, 64, 32, 32, 48, 48, k=1, seed=6)
    _compare(om, pm, batch, nn, mode='test')
    with pytest.raises(ValueError):
        pm.call(to_device_batch(batch, nn), 'bogus')


def test_forward_depth1024_config1_shape():
    # BASELINE config 1 (dragon_sss, depth 1024, 256^2 UV), one frame to keep the oracle quick
    om, pm = make_pair(depth=1024, uv=256, im=256, seed=7)
    batch, nn = O.synth_batch(1, 256, 256, 256, 256, 256, 256, k=1, seed=8, identity_warp=True)
    _compare(om, pm, batch, nn)


def test_no_obs_and_no_skip_base():
    om, pm = make_pair(depth=256, uv=64, im=64, seed=9, use_obs=False, skip_connect_base=False)
    batch, nn = O.synth_batch(1, 64, 64, 64, 64, 64, 64, k=1, seed=9)
    _compare(om, pm, batch, nn)


def test_relu_activation_branch_forward_and_train_step():
    """Config branch act = relu (negative slope 0): fused plan vs oracle, and one train step's gradients."""
    om, pm = make_pair(depth=256, uv=64, im=32, seed=13, act='relu')
    pm.build('cuda')
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=14)
    db = to_device_batch(batch, nn)
    with torch.no_grad():
        ref = om.call(batch, 'vali', nn_list=nn)
        got = pm.call(db, 'vali')
    assert rel_l2(got[3]['pred'].cpu(), ref[3]['pred']) <= TOL and rel_l2(got[0].cpu(), ref[0]) <= TOL
    pred, gt, kw, _ = pm(db, mode='train')
    loss = pm.compute_loss(pred, gt, keep_batch=True).sum() / 2
    pm.flat_params.grad = None
    loss.backward()
    po, go, _, _ = om.call(batch, 'train', nn_list=nn)
    lo = om.compute_loss(po, go, keep_batch=True).sum() / 2
    grads = torch.autograd.grad(lo, om.parameters())
    lo = lo.detach()
    assert abs(float(loss.detach()) - float(lo)) <= 1e-5 * abs(float(lo))
    it = iter(grads)
    worst = 0.0
    for c in pm._conv_layers():
        for name in ('dkernel', 'dbias'):
            g = next(it)
            worst = max(worst, float((getattr(c, name).cpu() - g).norm() / (g.norm() + 1e-12)))
    assert worst < 2e-3, worst


def test_layerwise_call_matches_fused_plan_and_oracle():
    """Model._call (reference structure, materialised concats, generic layer objects) ==
    fused plan == oracle."""
    om, pm = make_pair(depth=256, uv=64, im=64, seed=11)
    batch, nn = O.synth_batch(1, 64, 64, 64, 64, 64, 64, k=2, seed=12)
    db = to_device_batch(batch, nn)
    x = torch.cat((db[1], db[2], db[3]), 3)
    y_obs = [(db[9][:, i] - db[8][:, i]).contiguous() for i in range(2)]
    got = pm._call(x, y_obs)
    with torch.no_grad():
        ref = om._call(torch.cat((batch[1], batch[2], batch[3]), 3), [r - b for b, r in nn])
    assert rel_l2(got.cpu(), ref) <= TOL


def test_obs_override_path():
    om, pm = make_pair(depth=256, uv=64, im=64, seed=13)
    batch, nn = O.synth_batch(2, 64, 64, 64, 64, 64, 64, k=1, seed=14)
    with torch.no_grad():
        x = torch.cat((batch[1], batch[2], batch[3]), 3)
        _, feats = om._call(x, [nn[0][1] - nn[0][0]], return_feats=True)
        override = [f.mean(0, keepdim=True) for f in feats]          # nlt_test.py:118-126 style aggregate
        o_pred_c, _, _, o_vis = om.call(batch, 'test', obs_override=[f.expand(2, -1, -1, -1) for f in override],
                                        nn_list=nn)
    p_pred_c, _, _, p_vis = pm.call(to_device_batch(batch, nn), 'test', obs_override=[f.cuda() for f in override])
    assert rel_l2(p_vis['pred'].cpu(), o_vis['pred']) <= TOL


def test_golden_fixture_64():
    """Committed oracle outputs (tests/golden/make_forward_golden.py) -- guards the oracle AND
    the kernels against drifting together."""
    g = np.load(os.path.join(G, 'nlt_forward_64.npz'))
    om, pm = make_pair(depth=256, uv=64, im=32, seed=int(g['weight_seed']))
    batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=int(g['batch_seed']))
    p_pred_c, _, _, p_vis = pm.call(to_device_batch(batch, nn), 'train')
    assert rel_l2(p_vis['pred'].cpu(), g['pred']) <= TOL
    assert rel_l2(p_pred_c.cpu(), g['pred_camspc']) <= TOL


def test_nlt_test_orchestration_extract_feat_and_infer():
    """nlt/nlt_test.py:78-127 on the GPU: averaged observation features -> obs_override rendering."""
    from nlt_amd import nlt_test
    om, pm = make_pair(depth=256, uv=64, im=32, seed=15)
    train = [O.synth_batch(n, 64, 64, 32, 32, 32, 32, k=1, seed=60 + n) for n in (2, 3)]
    test_b, test_nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=70)
    with torch.no_grad():
        feats = [om._call(torch.cat((b[1], b[2], b[3]), 3), [b[5] - b[1]], return_feats=True)[1] for b, _ in train]
        ref_agg = [torch.cat([f[l] for f in feats], 0).mean(0, keepdim=True) for l in range(len(feats[0]))]
        ref = om.call(test_b, 'test', obs_override=[f.expand(2, -1, -1, -1) for f in ref_agg], nn_list=test_nn)[3]['pred']
    agg = nlt_test.extract_feat(pm, [to_device_batch(b, nn) for b, nn in train])
    for a, r in zip(agg, ref_agg):
        assert rel_l2(a.cpu(), r) <= TOL
    out = nlt_test.infer(pm, [to_device_batch(test_b, test_nn)], agg)
    assert rel_l2(out[0]['pred'].cpu(), ref) <= TOL


@pytest.mark.parametrize('mode', ['test', 'vali'])
def test_vis_batch_on_device_maps(tmp_path, mode):
    """nlt/models/nlt.py:207-272 with the maps in HBM: PNG bytes = oracle denormalize_float of the clipped maps, the PSNR
    in the metadata = the device kernel's float64 sums = the oracle's PSNR."""
    import json
    from nlt_amd.datasets.nlt import read_png
    from oracle import buffers as OB, metric as OM
    om, pm = make_pair(depth=256, uv=64, im=64, seed=21)
    batch, nn = O.synth_batch(2, 64, 64, 64, 64, 64, 64, k=1, seed=22)
    b = list(to_device_batch(batch, nn))
    b[0], b[7] = [b'cam0_light0', b'cam0_light1'], ['nn_a', 'nn_b']
    _, _, _, to_vis = pm.call(tuple(b), mode)
    pm.vis_batch(to_vis, str(tmp_path), mode)
    for i in range(2):
        for name in ('base', 'pred', 'nn') + (('gt',) if mode != 'test' else ()):
            want = OB.denormalize_float(np.clip(to_vis[name + '_camspc'][i].cpu().numpy(), 0, 1))
            np.testing.assert_array_equal(read_png(os.path.join(str(tmp_path), '%d_%s.png' % (i, name))), want)
        md = json.load(open(os.path.join(str(tmp_path), '%d_metadata.json' % i)))
        assert md['id'] == 'cam0_light%d' % i and md['nn_id'] == 'nn_' + 'ab'[i]
        if mode != 'test':
            gt, pred, base = (np.clip(to_vis[k][i].cpu().numpy(), 0, 1) for k in ('gt_camspc', 'pred_camspc', 'base_camspc'))
            assert md['pred_psnr'] == pytest.approx(OM.psnr(gt, pred), rel=1e-9)
            assert md['base_psnr'] == pytest.approx(OM.psnr(gt, base), rel=1e-9)
            assert pm.psnr(gt, pred) == pytest.approx(md['pred_psnr'], rel=1e-12)       # host arrays go up to the device
    link = pm.compile_batch_vis([str(tmp_path)], str(tmp_path / 'all'), mode)
    assert os.path.exists(str(tmp_path / 'all') + ('.apng' if mode == 'test' else '.html')) and link.startswith(str(tmp_path))


def test_infer_writes_batches_like_the_reference(tmp_path):
    from nlt_amd import nlt_test
    from nlt_amd.datasets.nlt import read_png
    from oracle import buffers as OB
    om, pm = make_pair(depth=256, uv=64, im=32, seed=15)
    train = [O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=62)]
    agg = nlt_test.extract_feat(pm, [to_device_batch(b, nn) for b, nn in train])
    tests = []
    for j in range(3):
        tb = list(to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=1, seed=70 + j)))
        tb[0], tb[7] = ['t%d_%d' % (j, i) for i in range(2)], ['n%d_%d' % (j, i) for i in range(2)]
        tests.append(tuple(tb))
    ref = nlt_test.infer(pm, tests, agg)
    for lanes, tag in ((1, 'a'), (2, 'b')):
        nlt_test.infer(pm, tests, agg, str(tmp_path / tag), lanes=lanes)
        for j in range(3):
            for i in range(2):
                want = OB.denormalize_float(np.clip(ref[j]['pred_camspc'][i].cpu().numpy(), 0, 1))
                np.testing.assert_array_equal(read_png(str(tmp_path / tag / ('batch%09d' % j) / ('%d_pred.png' % i))), want)
