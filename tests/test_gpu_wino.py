"""-m gpu: the Winograd F(2x2, 2x2) kernel (csrc/conv_wino.hip) against the oracle's Conv2D / Conv2DTranspose k2s1 'same' +
LeakyReLU: alone (channel-slice strides, ragged tiles, the in-register observation mean, its distance to FLOAT64 beside the
direct-sum kernel's), as backward-data with the mask / accumulate epilogue, after a one-launch weight refresh, and with every
eligible launch of Model.call / the train step routed to it."""
import numpy as np
import pytest
import torch

import nlt_amd
from nlt_amd import capi as C
from nlt_amd import trainvali
from oracle import nlt_oracle as O
from oracle import tf_ops as T
from gpu_util import rel_l2, make_pair, to_device_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cin,cout,tn,h,w,frames,kobs', [
    (16, 32, 32, 10, 20, 2, 2), (32, 64, 64, 8, 16, 1, 1), (64, 64, 32, 33, 47, 1, 3), (8, 32, 32, 9, 33, 2, 1),
    (256, 256, 64, 16, 16, 2, 1), (128, 128, 32, 64, 64, 1, 4), (64, 64, 64, 128, 128, 2, 1), (24, 96, 32, 17, 70, 1, 2)])
def test_conv_wino_vs_oracle(cin, cout, tn, h, w, frames, kobs):
    rng = np.random.default_rng(cin + cout + h + kobs)
    ld = cin + 8
    src = torch.from_numpy(rng.standard_normal((frames * kobs, h, w, ld)).astype(np.float32))
    wk = torch.from_numpy((rng.standard_normal((2, 2, cin, cout)) * (1.0 / np.sqrt(4 * cin))).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    with torch.no_grad():
        pre64 = T.conv2d_same(src[..., :cin].double().contiguous(), wk.double(), bias.double(), 1)
        ref = T.leaky_relu(pre64, 0.3)
    packed = C.pack_conv_wino_weights(C.CONV_K2S1, wk.cuda(), cin, cout, tn)
    fold = True                                                     # the running observation mean stays in registers in both forms
    out = torch.full((frames * kobs, h, w, cout + 4), float('nan'), device='cuda')        # ldo = cout + 4
    mean = torch.full((frames, h, w, 2 * cout), float('nan'), device='cuda')              # slice [cout, 2cout) of fm
    if fold:
        C.conv_wino_forward(C.CONV_K2S1, src.cuda(), ld, cin, frames, kobs, h, w, packed, bias.cuda(), cout, tn, out, cout + 4,
                            mean.view(-1)[cout:], 2 * cout, act=True, alpha=0.3)
    else:
        C.conv_wino_forward(C.CONV_K2S1, src.cuda(), ld, cin, frames * kobs, 1, h, w, packed, bias.cuda(), cout, tn, out, cout + 4,
                            None, 0, act=True, alpha=0.3)
    torch.cuda.synchronize()
    got = out[..., :cout].cpu()
    assert not torch.isnan(got).any() and torch.isnan(out[..., cout:]).all()            # nothing written outside the slice
    e_w = rel_l2(got, ref)
    # the direct-sum LDS-tiled kernel on the same inputs: Winograd may cost a small factor in rounding, not more
    e_d = None
    if cin % 16 == 0:
        p1 = C.pack_conv_tile_weights(C.CONV_K2S1, wk.cuda(), cin, cout, tn)
        o1 = torch.empty((frames * kobs, h, w, cout), device='cuda')
        C.conv_tile_forward(C.CONV_K2S1, src.cuda(), ld, cin, frames * kobs, 1, h, w, p1, bias.cuda(), cout, tn, o1, cout, None, 0)
        e_d = rel_l2(o1.cpu(), ref)
    assert e_w <= 2e-6 and (e_d is None or e_w <= 4 * e_d + 1e-7), (e_w, e_d)
    if fold:
        m = mean[..., cout:].cpu()
        assert torch.isnan(mean[..., :cout]).all() and not torch.isnan(m).any()
        assert rel_l2(m, ref.reshape(frames, kobs, h, w, cout).mean(1)) <= 2e-6
        mean2 = torch.empty((frames, h, w, cout), device='cuda')                         # mean only, no activation
        C.conv_wino_forward(C.CONV_K2S1, src.cuda(), ld, cin, frames, kobs, h, w, packed, bias.cuda(), cout, tn, None, 0, mean2, cout,
                            act=False)
        assert rel_l2(mean2.cpu(), pre64.reshape(frames, kobs, h, w, cout).mean(1)) <= 2e-6


@pytest.mark.parametrize('cin,cout,tn,h,w', [(16, 32, 32, 9, 33), (128, 128, 64, 32, 32), (32, 32, 32, 128, 128), (64, 64, 64, 10, 70)])
def test_conv_wino_transposed_vs_oracle(cin, cout, tn, h, w):
    """Forward Conv2DTranspose k2s1 'same' (the expanding blocks' second conv): (kh,kw,Cout,Cin) array, zero above / left."""
    rng = np.random.default_rng(cin + h)
    src = torch.from_numpy(rng.standard_normal((2, h, w, cin)).astype(np.float32))
    wk = torch.from_numpy((rng.standard_normal((2, 2, cout, cin)) * (1.0 / np.sqrt(4 * cin))).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    with torch.no_grad():
        ref = T.leaky_relu(T.conv2d_transpose_same(src.double(), wk.double(), bias.double(), 1), 0.3)
    packed = C.pack_conv_wino_weights(C.DECONV_K2S1, wk.cuda(), cin, cout, tn)
    out = torch.full((2, h, w, cout), float('nan'), device='cuda')
    C.conv_wino_forward(C.DECONV_K2S1, src.cuda(), cin, cin, 2, 1, h, w, packed, bias.cuda(), cout, tn, out, cout, None, 0)
    assert not torch.isnan(out).any() and rel_l2(out.cpu(), ref) <= 2e-6


@pytest.mark.parametrize('transpose_fwd', [False, True])
def test_conv_wino_backward_data_with_mask_and_accumulate(transpose_fwd):
    """Gradient w.r.t. input channels [lo, hi) of a forward Conv2D k2s1 (adjoint family DECONV_K2S1) / Conv2DTranspose k2s1
    (adjoint CONV_K2S1), read in place from the layer's own Keras array; epilogue: += target, x LeakyReLU'(mask)."""
    rng = np.random.default_rng(7 + transpose_fwd)
    cin_f, cout_f, lo, hi, h, w = 96, 64, 32, 96, 19, 45
    shape = (2, 2, cout_f, cin_f) if transpose_fwd else (2, 2, cin_f, cout_f)
    wk = torch.from_numpy((rng.standard_normal(shape) * 0.1).astype(np.float32))
    dpre = torch.from_numpy(rng.standard_normal((2, h, w, cout_f)).astype(np.float32))
    mask = torch.from_numpy(rng.standard_normal((2, h, w, hi - lo)).astype(np.float32))
    prev = torch.from_numpy(rng.standard_normal((2, h, w, hi - lo)).astype(np.float32))
    x = torch.zeros((2, h, w, cin_f), dtype=torch.float64, requires_grad=True)
    y = (T.conv2d_transpose_same if transpose_fwd else T.conv2d_same)(x, wk.double(), torch.zeros(cout_f, dtype=torch.float64), 1)
    (gx,) = torch.autograd.grad(y, x, dpre.double())
    ref = (prev.double() + gx[..., lo:hi]) * torch.where(mask > 0, 1.0, 0.3).double()
    adj = C.CONV_K2S1 if transpose_fwd else C.DECONV_K2S1
    for tn in (32, 64):
        packed = C.pack_conv_wino_weights(adj, wk.cuda(), cout_f, hi - lo, tn, full=cin_f, lo=lo)
        out = prev.cuda().clone()
        C.conv_wino_backward_data(adj, dpre.cuda(), cout_f, cout_f, 2, h, w, packed, hi - lo, tn, out, hi - lo, mask_src=mask.cuda(),
                                  ldm=hi - lo, mask_alpha=0.3, accumulate=True)
        assert rel_l2(out.cpu(), ref) <= 2e-6
        out2 = torch.full((2, h, w, hi - lo), float('nan'), device='cuda')                # plain: no mask, no accumulate
        C.conv_wino_backward_data(adj, dpre.cuda(), cout_f, cout_f, 2, h, w, packed, hi - lo, tn, out2, hi - lo)
        assert rel_l2(out2.cpu(), gx[..., lo:hi]) <= 2e-6


def _force_wino(pm, tn=32):
    """Every eligible launch of the plan on the Winograd kernel (a trial-style blanket hint), nothing else tuned."""
    pm.plan.autotune = False
    pm.plan._trial_wino = tn


@pytest.mark.parametrize('tn,k,uv', [(32, 3, 128), (64, 3, 128), (256 + 32, 3, 128), (256 + 64, 3, 128), (32, 1, 64), (64, 1, 64)])
def test_model_call_with_every_stride1_conv_on_the_winograd_kernel(tn, k, uv):
    om, pm = make_pair(depth=256, uv=uv, im=64, seed=17)
    batch, nn = O.synth_batch(2, uv, uv, 64, 64, 64, 64, k=k, seed=170 + k)
    with torch.no_grad():
        o_c, _, _, o_vis = om.call(batch, 'test', nn_list=nn)
    db = to_device_batch(batch, nn)
    pm.plan.two_streams = True
    _force_wino(pm, tn)
    ran = set()
    for _ in range(3):
        p_c, _, _, p_vis = pm.call(db, 'test')
        ran |= pm.plan._ran_wino
    torch.cuda.synchronize()
    assert any('.o.s1' in l for l in ran), ran
    if tn < 256:                                            # (+256 = "observations unfolded": a trial of the observation launches only)
        assert any('.q.s1' in l for l in ran), ran
        assert any(int(l.split('.')[0][1:]) > pm.plan.n_down for l in ran if l.endswith('.q.s1')), ran   # an expanding block's transposed conv
    assert rel_l2(p_vis['pred'].cpu(), o_vis['pred']) <= 2e-6 and rel_l2(p_c.cpu(), o_c) <= 2e-6


def test_plan_time_trials_may_choose_the_winograd_kernel_and_lanes_copy_the_choice():
    om, pm = make_pair(depth=256, uv=256, im=64, seed=19)
    batch, nn = O.synth_batch(2, 256, 256, 64, 64, 64, 64, k=2, seed=190)
    with torch.no_grad():
        o_c, _, _, o_vis = om.call(batch, 'test', nn_list=nn)
    db = to_device_batch(batch, nn)
    for _ in range(3):
        p_c, _, _, p_vis = pm.call(db, 'test')
    assert rel_l2(p_vis['pred'].cpu(), o_vis['pred']) <= 2e-6
    tuned = pm.plan.export_tuning()
    assert 'wino_hints' in tuned and not (set(tuned['wino_hints']) & set(tuned['lds_hints']))
    from nlt_amd.pipeline import RenderPipeline
    with RenderPipeline(pm, lanes=2) as pipe:
        outs = pipe.render([db, db, db], 'test')
    for o in outs:
        assert torch.equal(o[0], p_c)
    assert pipe._lanes[1].plan.wino_hints == pm.plan.wino_hints


def test_train_steps_with_the_winograd_kernel_in_forward_and_backward_data():
    """Forward stride-1 convs AND their backward-data launches on the Winograd kernel: per-step loss and the weights after three
    Adam steps against the oracle's autograd; the packed G g G^T fragments follow the optimizer through the one-launch refresh."""
    om, pm = make_pair(depth=256, uv=128, im=64, loss='l2', seed=23)
    pm.build('cuda')
    batch, nn = O.synth_batch(2, 128, 128, 64, 64, 64, 64, k=1, seed=230)
    db = to_device_batch(batch, nn)
    _force_wino(pm, 32)
    pm.plan.tune_backward = False
    opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
    oopt = O.KerasAdamAMSGrad(om.parameters(), 1e-3)
    ran = set()
    for step in range(3):
        lo, _ = O.train_step(om, oopt, batch, global_bs=2, nn_list=nn)
        lp = float(trainvali.distributed_train_step(pm, db, opt, 2)[0])
        ran |= pm.plan._ran_wino
        assert abs(lp - float(lo)) <= 2e-5 * max(1.0, abs(float(lo))), (step, lp, float(lo))
    assert any('dgrad' in l for l in ran) and any(l.endswith('.o.s1') for l in ran), ran
    worst = max(float((po_.detach() - c.kernel.cpu()).abs().max()) for po_, c in zip(om.parameters()[::2], pm._conv_layers()))
    assert worst < 2e-4, worst
