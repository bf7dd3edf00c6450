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
#   every single kernel / bias                                       <= GRAD_TOL_TENSOR
# For the released LeakyReLU(0.3) the per-tensor comparison is MASK-CONDITIONED: the float64 oracle takes, at every
# activation, the branch the HIP forward took (`OracleModel.act_masks` <- `gpu_util.hip_activation_masks`).  LeakyReLU's
# derivative is discontinuous, so two correct forwards that round a pre-activation to opposite sides of zero differ by one
# whole texel's contribution to a gradient (r02: 2.4e-3 on a 256-channel kernel of a 32^2-texel level) -- that is a
# property of the function, not an error of either side, and with the branches shared what is left is accumulation error.
# (alpha = 1 needs no conditioning: the activation is the identity.)
GRAD_TOL_FLAT = 1e-5
GRAD_TOL_TENSOR = 1e-5
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


def _forward_vs_oracle(name, depth, uv, cam, n, k, identity_warp, seed, tol=TOL, **product_only):
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    om, pm = make_pair(depth=depth, uv=uv, im=cam, seed=seed, **product_only)
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
    assert e_uv <= tol and e_cam <= tol and e_base <= 1e-6, (e_uv, e_cam, e_base)
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


@pytest.mark.parametrize('n', [1, 2, 4])
def test_config3_1024_k4_random_warp(n):
    """BASELINE config 3 (the bench workload): depth 256, 1024^2 UV, k = 4 observation maps, 512^2 random fg/bg warp.
    n = 4 is the bench's exact shape (plan-time choices depend on size)."""
    _forward_vs_oracle('config3_1024_k4_n%d' % n, 256, 1024, 512, n, 4, False, seed=3 + n)


def test_config1_depth1024_256_4_frames():
    """BASELINE config 1's full shape: dragon_sss, depth 1024 (18 query / 9 obs layers), 4 frames, 256^2 UV, k = 1."""
    _forward_vs_oracle('config1_d1024_256_k1_n4', 1024, 256, 256, 4, 1, True, seed=1)


@pytest.mark.parametrize('n', [1, 2])
def test_config5_2048_fp32_and_bf16(n):
    """BASELINE config 5's size: depth 256, 2048^2 UV, k = 1, 512^2 random fg/bg warp, 1 and 2 frames (the bench's
    `config5_2048_bf16` sub-line runs 2).  The plan-time choices (wave tile, split-K, LDS-tiled vs register-tiled kernel)
    depend on size, so the launch configurations the sub-line is measured with are exactly the ones compared here.
      fp32 plan   vs the fp32 oracle: rendered texels <= 1e-4 rel-L2, UV gather indices bit-exact;
      bf16 middle vs OracleModel.set_precision('bf16') (same operands rounded to bf16 at the same places, fp32
      accumulation): <= 5e-3 on the map leaving the bf16 region, <= 1e-4 on the rendered texels (DESIGN.md section 3;
      tests/test_gpu_bf16.py states the same bars at 128^2 / 256^2); the distance to the fp32 oracle is recorded."""
    import nlt_amd
    from nlt_amd.models import get_model_class
    uv, cam = 2048, 512
    pm, db, o_vis = _forward_vs_oracle('config5_2048_k1_n%d_fp32' % n, 256, uv, cam, n, 1, False, seed=50 + n)
    # same weights, same batch, bf16 middle
    om = O.OracleModel(depth=256, uvh=uv, uvw=uv, imh=cam, imw=cam, seed=50 + n)
    batch, nn = O.synth_batch(n, uv, uv, cam, cam, cam, cam, k=1, seed=150 + n)
    pb = get_model_class('nlt')(nlt_amd.make_config(depth=256, uvh=uv, uvw=uv, imh=cam, imw=cam, precision='bf16'))
    pb.load_weights(om.numpy_weights())
    pb.register_trainable()
    om.set_precision('bf16')
    outs = []
    with torch.no_grad():
        om._call(torch.cat((batch[1], batch[2], batch[3]), 3), [r - b for b, r in nn], layer_outputs=outs)
        o16_c, _, _, o16 = om.call(batch, 'test', nn_list=nn)
    for _ in range(3):
        p_c, _, _, p_vis = pb.call(db, 'test', want_indices=True)
    torch.cuda.synchronize()
    bufs = next(iter(pb.plan._bufs.values()))
    D = pb.plan.n_down
    assert bufs['fm'][3].dtype == torch.bfloat16 and bufs['dec'][D - 3].dtype == torch.float32
    e_region = rel_l2(bufs['dec'][D - 3].cpu(), outs[D + 1 + D - 3])
    e16_uv, e16_cam = rel_l2(p_vis['pred'].cpu(), o16['pred']), rel_l2(p_c.cpu(), o16_c)
    e32_uv = rel_l2(p_vis['pred'].cpu(), o_vis['pred'])
    _dump('config5_2048_k1_n%d_bf16' % n, {'rel_l2_region_out_vs_bf16_oracle': e_region, 'rel_l2_pred_uv_vs_bf16_oracle': e16_uv,
                                           'rel_l2_pred_camspc_vs_bf16_oracle': e16_cam, 'rel_l2_pred_uv_vs_fp32_oracle': e32_uv})
    assert e_region <= 5e-3 and e16_uv <= TOL and e16_cam <= TOL, (e_region, e16_uv, e16_cam)
    fx, fy, inside = T.resampler_indices(o16['warp_px'].numpy(), uv, uv)
    idx = p_vis['uv_indices'].cpu().numpy()
    np.testing.assert_array_equal(idx[..., 0], fx)
    np.testing.assert_array_equal(idx[..., 1], fy)
    np.testing.assert_array_equal(idx[..., 2], inside.astype(np.int32))


@pytest.mark.parametrize('precision', ['f32x3', 'f32x3_9'])
@pytest.mark.parametrize('cfg', [1, 2, 3])
def test_three_term_split_forward_at_configs_1_2_3(cfg, precision):
    """precision = f32x3 (6 term products) / f32x3_9 (all 9) at BASELINE configs 1-3 against the fp32 oracle: rendered texels
    <= 1e-6 rel-L2 (the bar the r02 review set for this mode; the native fp32 path measures 1e-7), UV gather indices bit-exact."""
    depth, uv, cam, n, k, ident = {1: (1024, 256, 256, 4, 1, True), 2: (256, 512, 512, 4, 1, True), 3: (256, 1024, 512, 2, 4, False)}[cfg]
    pm, _, _ = _forward_vs_oracle('config%d_%s' % (cfg, precision), depth, uv, cam, n, k, ident, seed=cfg, tol=1e-6, precision=precision)
    _dump('config%d_%s_launches_on_the_split_kernel' % (cfg, precision), sorted(pm.plan.lds_hints))
    _dump('config%d_%s_launches_on_the_winograd_kernel' % (cfg, precision), sorted(pm.plan.wino_hints))
    if cfg == 3:
        assert pm.plan.lds_hints or pm.plan.wino_hints, "no launch left the native kernels at the size it is benchmarked at"


def test_headline_precision_at_the_bench_shape_and_at_config_5():
    """The bench's headline plan (precision = f32x3_9: nine exact bf16 term products, fp32 accumulate, + whatever the plan-time
    trials gave to the Winograd kernel) at the bench's EXACT shape (config 3, 4 frames) and at BASELINE config 5's size (2048^2,
    k = 1, 2 frames): rendered texels <= 1e-6 rel-L2 against the fp32 oracle, UV gather indices bit-exact (VERDICT r03's
    conditions for this precision to carry the headline)."""
    _forward_vs_oracle('config3_1024_k4_n4_f32x3_9', 256, 1024, 512, 4, 4, False, seed=7, tol=1e-6, precision='f32x3_9')
    _forward_vs_oracle('config5_2048_k1_n2_f32x3_9', 256, 2048, 512, 2, 1, False, seed=52, tol=1e-6, precision='f32x3_9')


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


def _oracle_grads(loss, uv, cam, n, dtype, batch, nn, alpha=None, masks=None):
    om = O.OracleModel(depth=256, uvh=uv, uvw=uv, imh=cam, imw=cam, loss=loss, seed=41, dtype=dtype)
    if alpha is not None:
        om.alpha = alpha
    om.act_masks = masks
    b = tuple(t.to(dtype) if torch.is_tensor(t) else t for t in batch)
    nnl = [(a.to(dtype), c.to(dtype)) for a, c in nn]
    po, go, _, _ = om.call(b, 'train', nn_list=nnl)
    lo = om.compute_loss(po, go, keep_batch=True).sum() / n
    grads = [g.double() for g in torch.autograd.grad(lo, om.parameters())]
    return float(lo.detach()), grads


def _per_tensor(pm, grads):
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
    return (num / den) ** 0.5, sorted(zip(errs, names), reverse=True)[:8]


@pytest.mark.parametrize('loss,alpha,n,precision', [('l2', 0.3, 1, 'fp32'), ('barron', 0.3, 1, 'fp32'), ('l2', 1.0, 1, 'fp32'),
                                                    ('barron', 1.0, 1, 'fp32'), ('l2', 0.3, 4, 'fp32'), ('l2', 0.3, 1, 'f32x3'),
                                                    ('barron', 0.3, 1, 'f32x3_9')])
def test_config4_train_step_1024_per_tensor_gradients(loss, alpha, n, precision):
    """BASELINE config 4's per-GPU shape (1024^2 UV, 512^2 camera, k = 1; n = 4 frames is exactly what bench.py trains):
    loss and EVERY weight / bias gradient of one train step against the oracle's float64 autograd
    (nlt/trainvali.py:272-281), every tensor <= 1e-5.

    alpha = 0.3 is the released LeakyReLU: compared against the float64 oracle evaluated on the HIP forward's own
    activation branches (see GRAD_TOL_TENSOR above); the unconditioned float64 oracle still has to give the same loss and
    the same flat bucket to 1e-5.  alpha = 1.0 is the kink-free twin and needs no conditioning.
    precision = f32x3 / f32x3_9: the train FORWARD's LDS-tiled encoder convs on the three-term bf16 split (csrc/conv_tile3.hip;
    the backward stays native fp32) -- same bars."""
    from gpu_util import hip_activation_masks
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    uv, cam = 1024, 512
    om32, pm = make_pair(depth=256, uv=uv, im=cam, loss=loss, seed=41, **({} if precision == 'fp32' else {'precision': precision}))
    _set_alpha(om32, pm, alpha)
    pm.build('cuda')
    batch, nn = O.synth_batch(n, uv, uv, cam, cam, cam, cam, k=1, seed=141)
    lo, grads = _oracle_grads(loss, uv, cam, n, torch.float64, batch, nn, alpha)

    db = to_device_batch(batch, nn)
    recs = []
    for rep in range(2):                                    # second pass: recorded launch tape / tuned plan
        pred, gt, kw, _ = pm(db, mode='train')
        lp = pm.compute_loss(pred, gt, keep_batch=True).sum() / n
        pm.flat_params.grad = None
        lp.backward()
        torch.cuda.synchronize()
        assert abs(float(lp.detach()) - lo) <= 1e-5 * abs(lo), (loss, float(lp.detach()), lo)
        flat, worst = _per_tensor(pm, grads)
        rec = {'loss_hip': float(lp.detach()), 'loss_oracle_f64': lo, 'flat_rel': flat, 'worst_unconditioned': worst}
        if alpha != 1.0:
            masks = hip_activation_masks(pm)
            lo_m, grads_m = _oracle_grads(loss, uv, cam, n, torch.float64, batch, nn, alpha, masks=masks)
            flat_m, worst_m = _per_tensor(pm, grads_m)
            rec.update({'loss_oracle_f64_hip_masks': lo_m, 'flat_rel_hip_masks': flat_m, 'worst_hip_masks': worst_m})
            assert abs(float(lp.detach()) - lo_m) <= 1e-5 * abs(lo_m)
        recs.append(rec)
        if n > 1:
            break                                           # (the 4-frame float64 oracle pass is the expensive part)
    if precision != 'fp32':
        recs[-1]['launches_on_the_split_kernel'] = sorted(pm.plan.lds_hints)
    _dump('config4_train_1024_%s_alpha%g_n%d%s' % (loss, alpha, n, '' if precision == 'fp32' else '_' + precision), recs[-1])
    for r in recs:
        assert r['flat_rel'] <= GRAD_TOL_FLAT, (loss, r['flat_rel'])
        worst = r['worst_unconditioned'] if alpha == 1.0 else r['worst_hip_masks']
        assert worst[0][0] <= GRAD_TOL_TENSOR, (loss, alpha, worst[:4])
        if alpha != 1.0:
            assert r['flat_rel_hip_masks'] <= GRAD_TOL_FLAT
