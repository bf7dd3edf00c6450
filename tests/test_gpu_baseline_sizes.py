"""-m gpu: HIP path vs the CPU oracle AT THE BASELINE.json CONFIG SIZES (not plan-vs-plan).

Every comparison below is `libnlt_hip.so through Model.call / the train step` against `oracle.OracleModel` on the
same seeded inputs at the full shapes of BASELINE configs 1-4: the mid-network kernels (LDS-tiled convs, register-
tiled MFMA convs, split-K, the plan-time choices that depend on size) are exercised at the sizes they are
benchmarked at.  The oracle needs ~0.5 s per 1024^2, k = 4 frame on 32 host threads (bench.py's own cpu_baseline).

Bars: rendered texels <= 1e-4 rel-L2; integer UV gather indices bit-exact; train-step loss and gradients against
the oracle's float64 autograd, PER TENSOR (every kernel and bias of both nets), bounds GRAD_TOL_* below.
"""
import json
import os

import numpy as np
import pytest
import torch

import nlt_amd
from oracle import nlt_oracle as O
from oracle import tf_ops as T
from gpu_util import rel_l2, make_pair, to_device_batch

pytestmark = pytest.mark.gpu
TOL = 1e-4
# Bounds on train-step gradients vs the FLOAT64 oracle:
#   flat gradient bucket (what the all-reduce and Adam see)          <= 1e-5 rel-L2 (measured 1e-7)
#   every single kernel / bias, kink-free network (alpha = 1)        <= GRAD_TOL_SMOOTH (measured 1e-6; the fp32 torch-CPU
#       oracle itself sits at 1.4e-5 .. 1.9e-5 from float64 on its worst tensor)
#   every single kernel / bias, released LeakyReLU(0.3)              <= max(GRAD_TOL_KINK, 3 x what the fp32 torch-CPU
#       oracle itself is away from float64 on its worst tensor): derivative-mask flips of the few texels whose
#       pre-activation is within fp32 rounding of zero.  Whether a flip lands on a tensor is chance; at a 32^2-texel deep
#       layer under the Barron loss ONE flipped texel moved a 256-channel kernel gradient by 2.4e-3 (measured, r02).
GRAD_TOL_FLAT = 1e-5
GRAD_TOL_SMOOTH = 1e-5
GRAD_TOL_KINK = 5e-3
DUMP = os.environ.get('NLT_PARITY_DUMP')


def _dump(name, rec):
    if DUMP:
        os.makedirs(os.path.dirname(DUMP) or '.', exist_ok=True)
        try:
            with open(DUMP) as f:
                d = json.load(f)
        except (OSError, ValueError):
            d = {}
        d[name] = rec
        with open(DUMP, 'w') as f:
            json.dump(d, f, indent=1)


def _forward_vs_oracle(name, depth, uv, cam, n, k, identity_warp, seed):
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    om, pm = make_pair(depth=depth, uv=uv, im=cam, seed=seed)
    batch, nn = O.synth_batch(n, uv, uv, cam, cam, cam, cam, k=k, seed=seed + 100, identity_warp=identity_warp)
    with torch.no_grad():
        o_pred_c, _, _, o_vis = om.call(batch, 'test', nn_list=nn)
    db = to_device_batch(batch, nn)
    for _ in range(3):                                      # plan-time autotune, launch tape record, then a replayed step
        p_pred_c, _, _, p_vis = pm.call(db, 'test', want_indices=True)
    torch.cuda.synchronize()
    e_uv = rel_l2(p_vis['pred'].cpu(), o_vis['pred'])
    e_cam = rel_l2(p_pred_c.cpu(), o_pred_c)
    e_base = rel_l2(p_vis['base_camspc'].cpu(), o_vis['base_camspc'])
    _dump(name, {'rel_l2_pred_uv': e_uv, 'rel_l2_pred_camspc': e_cam, 'rel_l2_base_camspc': e_base})
    assert e_uv <= TOL and e_cam <= TOL and e_base <= 1e-6, (e_uv, e_cam, e_base)
    fx, fy, inside = T.resampler_indices(o_vis['warp_px'].numpy(), uv, uv)
    idx = p_vis['uv_indices'].cpu().numpy()
    np.testing.assert_array_equal(idx[..., 0], fx)
    np.testing.assert_array_equal(idx[..., 1], fy)
    np.testing.assert_array_equal(idx[..., 2], inside.astype(np.int32))
    return pm, db, o_vis


def test_config2_512_relight_only_4_frames():
    """BASELINE config 2: dragon_specular relight-only, depth 256, 4 frames, 512^2 UV, identity warp, k = 1."""
    pm, db, o_vis = _forward_vs_oracle('config2_512_k1_n4', 256, 512, 512, 4, 1, True, seed=2)
    # identity warp: the camera-space prediction IS the UV prediction (texel (0,0) zeroed in both)
    _, _, _, vis = pm.call(db, 'test')
    np.testing.assert_array_equal(vis['pred_camspc'].cpu().numpy(), vis['pred'].cpu().numpy())


@pytest.mark.parametrize('n', [1, 2])
def test_config3_1024_k4_random_warp(n):
    """BASELINE config 3 (the bench workload): depth 256, 1024^2 UV, k = 4 observation maps, 512^2 random fg/bg warp."""
    _forward_vs_oracle('config3_1024_k4_n%d' % n, 256, 1024, 512, n, 4, False, seed=3 + n)


def test_config1_depth1024_256_4_frames():
    """BASELINE config 1's full shape: dragon_sss, depth 1024 (18 query / 9 obs layers), 4 frames, 256^2 UV, k = 1."""
    _forward_vs_oracle('config1_d1024_256_k1_n4', 1024, 256, 256, 4, 1, True, seed=1)


def _set_alpha(om, pm, alpha):
    """Same negative slope on both sides (alpha = 1: LeakyReLU becomes the identity -- a kink-free network)."""
    from nlt_amd.networks.elements import Act, Sequential
    om.alpha = alpha
    for net in pm.net.values():
        for blk in net.layers:
            if isinstance(blk, Sequential):
                for l in blk.layers:
                    if isinstance(l, Act):
                        l.alpha = alpha


def _oracle_grads(loss, uv, cam, n, dtype, batch, nn, alpha=None):
    om = O.OracleModel(depth=256, uvh=uv, uvw=uv, imh=cam, imw=cam, loss=loss, seed=41, dtype=dtype)
    if alpha is not None:
        om.alpha = alpha
    b = tuple(t.to(dtype) if torch.is_tensor(t) else t for t in batch)
    nnl = [(a.to(dtype), c.to(dtype)) for a, c in nn]
    po, go, _, _ = om.call(b, 'train', nn_list=nnl)
    lo = om.compute_loss(po, go, keep_batch=True).sum() / n
    grads = [g.double() for g in torch.autograd.grad(lo, om.parameters())]
    return float(lo.detach()), grads


@pytest.mark.parametrize('alpha', [0.3, 1.0])
@pytest.mark.parametrize('loss', ['l2', 'barron'])
def test_config4_train_step_1024_per_tensor_gradients(loss, alpha):
    """BASELINE config 4's per-GPU shape (1024^2 UV, 512^2 camera, k = 1; one frame): loss and EVERY weight / bias
    gradient of one train step against the oracle's float64 autograd (nlt/trainvali.py:272-281).

    alpha = 0.3 is the released LeakyReLU.  Its derivative is discontinuous, so ANY fp32 forward flips the mask of
    the few texels whose pre-activation is within fp32 rounding of zero; the fp32 torch-CPU oracle itself is
    0.4-1.0e-3 away from the float64 oracle on single tensors (asserted below, beside the HIP numbers).  alpha = 1.0
    turns the activation into the identity: no kinks, and every tensor has to agree to GRAD_TOL_SMOOTH."""
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    uv, cam, n = 1024, 512, 1
    om32, pm = make_pair(depth=256, uv=uv, im=cam, loss=loss, seed=41)
    _set_alpha(om32, pm, alpha)
    pm.build('cuda')
    batch, nn = O.synth_batch(n, uv, uv, cam, cam, cam, cam, k=1, seed=141)
    lo, grads = _oracle_grads(loss, uv, cam, n, torch.float64, batch, nn, alpha)
    _, grads32 = _oracle_grads(loss, uv, cam, n, torch.float32, batch, nn, alpha)
    o32_worst = max(float((a - b).norm() / b.norm()) for a, b in zip(grads32, grads))

    db = to_device_batch(batch, nn)
    recs = []
    for rep in range(2):                                    # second pass: recorded launch tape / tuned plan
        pred, gt, kw, _ = pm(db, mode='train')
        lp = pm.compute_loss(pred, gt, keep_batch=True).sum() / n
        pm.flat_params.grad = None
        lp.backward()
        torch.cuda.synchronize()
        assert abs(float(lp.detach()) - lo) <= 1e-5 * abs(lo), (loss, float(lp.detach()), lo)
        it = iter(grads)
        names, errs = [], []
        num = den = 0.0
        for li, c in enumerate(pm._conv_layers()):
            for nm in ('dkernel', 'dbias'):
                g = next(it)
                got = getattr(c, nm).detach().cpu().double()
                d = float((got - g).norm())
                r = float(g.norm())
                num += d * d; den += r * r
                names.append('conv%d.%s%s' % (li, nm, tuple(g.shape)))
                errs.append(d / max(r, 1e-300))
        recs.append({'loss_hip': float(lp.detach()), 'loss_oracle_f64': lo, 'flat_rel': (num / den) ** 0.5,
                     'fp32_oracle_worst_tensor_vs_f64': o32_worst,
                     'worst': sorted(zip(errs, names), reverse=True)[:8]})
    _dump('config4_train_1024_%s_alpha%g' % (loss, alpha), recs[-1])
    for r in recs:
        assert r['flat_rel'] <= GRAD_TOL_FLAT, (loss, r['flat_rel'])
        if alpha == 1.0:
            assert r['worst'][0][0] <= GRAD_TOL_SMOOTH, (loss, r['worst'][:4])
        else:
            assert r['worst'][0][0] <= max(GRAD_TOL_KINK, 3 * o32_worst), (loss, o32_worst, r['worst'][:4])
