"""-m gpu: the LDS-tiled encoder conv (csrc/conv_tile.hip) against the oracle's Conv2D 'same' + LeakyReLU, alone
(channel-slice strides, partial tiles, observation mean) and with every eligible launch of Model.call routed to it."""
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from oracle import nlt_oracle as O
from oracle import tf_ops as T
from gpu_util import rel_l2, make_pair, to_device_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode,cin,cout,tn,h,w,frames,kobs', [
    (C.CONV_K2S1, 16, 32, 32, 10, 20, 2, 2), (C.CONV_K2S1, 32, 64, 64, 8, 16, 1, 1), (C.CONV_K2S1, 64, 64, 32, 33, 47, 1, 3),
    (C.CONV_K2S1, 256, 256, 64, 16, 16, 2, 2), (C.CONV_K2S1, 128, 128, 64, 64, 64, 1, 4),
    (C.CONV_K2S2, 16, 32, 32, 20, 36, 2, 2), (C.CONV_K2S2, 32, 128, 64, 16, 32, 1, 1), (C.CONV_K2S2, 64, 64, 64, 66, 94, 1, 2),
    (C.CONV_K2S2, 512, 256, 64, 32, 32, 1, 1), (C.CONV_K2S2, 32, 32, 32, 256, 256, 2, 1)])
def test_conv_tile_vs_oracle(mode, cin, cout, tn, h, w, frames, kobs):
    rng = np.random.default_rng(cin + cout + h + kobs)
    ld = cin + 8
    src = torch.from_numpy(rng.standard_normal((frames * kobs, h, w, ld)).astype(np.float32))
    wk = torch.from_numpy((rng.standard_normal((2, 2, cin, cout)) * (1.0 / np.sqrt(4 * cin))).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    stride = 2 if mode == C.CONV_K2S2 else 1
    with torch.no_grad():
        ref = T.leaky_relu(T.conv2d_same(src[..., :cin].contiguous(), wk, bias, stride), 0.3)
    oh, ow = ref.shape[1:3]
    packed = C.pack_conv_tile_weights(mode, wk.cuda(), cin, cout, tn)
    out = torch.full((frames * kobs, oh, ow, cout + 4), float('nan'), device='cuda')      # ldo = cout + 4
    mean = torch.full((frames, oh, ow, 2 * cout), float('nan'), device='cuda')            # slice [cout, 2cout) of fm
    C.conv_tile_forward(mode, src.cuda(), ld, cin, frames, kobs, h, w, packed, bias.cuda(), cout, tn, out, cout + 4,
                        mean.view(-1)[cout:], 2 * cout, act=True, alpha=0.3)
    torch.cuda.synchronize()
    got = out[..., :cout].cpu()
    assert not torch.isnan(got).any() and torch.isnan(out[..., cout:]).all()            # nothing written outside the slice
    assert rel_l2(got, ref) <= 1e-5
    m = mean[..., cout:].cpu()
    assert torch.isnan(mean[..., :cout]).all() and not torch.isnan(m).any()
    assert rel_l2(m, ref.reshape(frames, kobs, oh, ow, cout).mean(1)) <= 1e-5
    # mean only (no per-frame output) and no activation
    mean2 = torch.empty((frames, oh, ow, cout), device='cuda')
    C.conv_tile_forward(mode, src.cuda(), ld, cin, frames, kobs, h, w, packed, bias.cuda(), cout, tn, None, 0, mean2, cout,
                        act=False)
    with torch.no_grad():
        ref2 = T.conv2d_same(src[..., :cin].contiguous(), wk, bias, stride).reshape(frames, kobs, oh, ow, cout).mean(1)
    assert rel_l2(mean2.cpu(), ref2) <= 1e-5
    # the observations as plain frames, no mean (the kernel's mean-free instantiation: query path / unfolded launches)
    out3 = torch.full((frames * kobs, oh, ow, cout), float('nan'), device='cuda')
    C.conv_tile_forward(mode, src.cuda(), ld, cin, frames * kobs, 1, h, w, packed, bias.cuda(), cout, tn, out3, cout, None, 0,
                        act=True, alpha=0.3)
    assert rel_l2(out3.cpu(), ref) <= 1e-5


@pytest.mark.parametrize('cin,h,w,frames,kobs', [(32, 10, 20, 2, 2), (16, 8, 16, 1, 1), (32, 33, 47, 1, 3), (32, 256, 256, 1, 4), (16, 17, 40, 2, 2)])
def test_conv_c32_vs_oracle(cin, h, w, frames, kobs):
    """csrc/conv_c32.hip: the narrow stride-1 conv (cin 16 | 32 -> 32) with LDS-resident weights and a frame per stage, against
    the oracle's Conv2D k2s1 'same' + LeakyReLU: channel-slice strides, ragged tiles, the in-register observation mean, mean only."""
    cout = 32
    rng = np.random.default_rng(cin + h + kobs)
    ld = cin + 8
    src = torch.from_numpy(rng.standard_normal((frames * kobs, h, w, ld)).astype(np.float32))
    wk = torch.from_numpy((rng.standard_normal((2, 2, cin, cout)) * (1.0 / np.sqrt(4 * cin))).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    with torch.no_grad():
        pre = T.conv2d_same(src[..., :cin].contiguous(), wk, bias, 1)
        ref = T.leaky_relu(pre, 0.3)
    assert C.conv_c32_supported(C.CONV_K2S1, cin, cout) and not C.conv_c32_supported(C.CONV_K2S1, 64, 32)
    packed = C.pack_conv_tile_weights(C.CONV_K2S1, wk.cuda(), cin, cout, 32)
    out = torch.full((frames * kobs, h, w, cout + 4), float('nan'), device='cuda')
    mean = torch.full((frames, h, w, 2 * cout), float('nan'), device='cuda')
    C.conv_c32_forward(C.CONV_K2S1, src.cuda(), ld, cin, frames, kobs, h, w, packed, bias.cuda(), cout, out, cout + 4,
                       mean.view(-1)[cout:], 2 * cout, act=True, alpha=0.3)
    got = out[..., :cout].cpu()
    assert not torch.isnan(got).any() and torch.isnan(out[..., cout:]).all()
    assert rel_l2(got, ref) <= 1e-5
    m = mean[..., cout:].cpu()
    assert torch.isnan(mean[..., :cout]).all() and not torch.isnan(m).any()
    assert rel_l2(m, ref.reshape(frames, kobs, h, w, cout).mean(1)) <= 1e-5
    mean2 = torch.empty((frames, h, w, cout), device='cuda')
    C.conv_c32_forward(C.CONV_K2S1, src.cuda(), ld, cin, frames, kobs, h, w, packed, bias.cuda(), cout, None, 0, mean2, cout, act=False)
    assert rel_l2(mean2.cpu(), pre.reshape(frames, kobs, h, w, cout).mean(1)) <= 1e-5
    out3 = torch.full((frames * kobs, h, w, cout), float('nan'), device='cuda')
    C.conv_c32_forward(C.CONV_K2S1, src.cuda(), ld, cin, frames * kobs, 1, h, w, packed, bias.cuda(), cout, out3, cout, None, 0)
    assert rel_l2(out3.cpu(), ref) <= 1e-5


@pytest.mark.parametrize('nprod', [6, 9])
@pytest.mark.parametrize('mode,cin,cout,tn,h,w,frames,kobs', [
    (C.CONV_K2S1, 16, 32, 32, 10, 20, 2, 2), (C.CONV_K2S1, 32, 64, 64, 8, 16, 1, 1), (C.CONV_K2S1, 64, 64, 32, 33, 47, 1, 3),
    (C.CONV_K2S1, 256, 256, 64, 16, 16, 2, 2), (C.CONV_K2S1, 128, 128, 64, 64, 64, 1, 4),
    (C.CONV_K2S2, 16, 32, 32, 20, 36, 2, 2), (C.CONV_K2S2, 32, 128, 64, 16, 32, 1, 1), (C.CONV_K2S2, 64, 64, 64, 66, 94, 1, 2),
    (C.CONV_K2S2, 512, 256, 64, 32, 32, 1, 1)])
def test_conv_tile3_three_term_bf16_split_vs_float64(nprod, mode, cin, cout, tn, h, w, frames, kobs):
    """csrc/conv_tile3.hip (precision = f32x3): fp32 operands as three bf16 terms on v_mfma_f32_16x16x32_bf16, against the
    conv evaluated in FLOAT64 -- beside the native fp32 MFMA kernel's distance to the same float64 result.  Bars: both forms have
    to be as close to float64 as the native fp32 kernel (x 1.5: all three only round in the fp32 accumulation -- the
    three products the 6-product form drops are 6e-9 relative, r03 measurement: the two forms agree to 3 digits with each
    other and sit slightly BELOW the native kernel's error).  Slice strides, partial tiles, the observation mean and the
    no-activation / mean-only form as in the fp32 test above."""
    rng = np.random.default_rng(cin + cout + h + kobs)
    ld = cin + 8
    src = torch.from_numpy(rng.standard_normal((frames * kobs, h, w, ld)).astype(np.float32))
    wk = torch.from_numpy((rng.standard_normal((2, 2, cin, cout)) * (1.0 / np.sqrt(4 * cin))).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    stride = 2 if mode == C.CONV_K2S2 else 1
    with torch.no_grad():
        pre64 = T.conv2d_same(src[..., :cin].double().contiguous(), wk.double(), bias.double(), stride)
        ref = T.leaky_relu(pre64, 0.3)
    oh, ow = ref.shape[1:3]
    E = lambda: torch.full((frames * kobs, oh, ow, cout + 4), float('nan'), device='cuda')
    out3, out1 = E(), E()
    mean = torch.full((frames, oh, ow, 2 * cout), float('nan'), device='cuda')
    C.conv_tile3_forward(mode, src.cuda(), ld, cin, frames, kobs, h, w, C.pack_conv_tile3_weights(mode, wk.cuda(), cin, cout, tn),
                         bias.cuda(), cout, tn, out3, cout + 4, mean.view(-1)[cout:], 2 * cout, act=True, alpha=0.3, nprod=nprod)
    C.conv_tile_forward(mode, src.cuda(), ld, cin, frames, kobs, h, w, C.pack_conv_tile_weights(mode, wk.cuda(), cin, cout, tn),
                        bias.cuda(), cout, tn, out1, cout + 4, None, 0, act=True, alpha=0.3)
    torch.cuda.synchronize()
    got = out3[..., :cout].cpu()
    assert not torch.isnan(got).any() and torch.isnan(out3[..., cout:]).all()
    e3, e1 = rel_l2(got, ref), rel_l2(out1[..., :cout].cpu(), ref)
    print("f32x3 nprod=%d: rel-L2 vs float64 %.2e (native fp32 MFMA kernel %.2e)" % (nprod, e3, e1))
    assert e3 <= 1.5 * e1 + (1e-9 if nprod == 9 else 3e-8), (e3, e1)
    m = mean[..., cout:].cpu()
    assert torch.isnan(mean[..., :cout]).all() and rel_l2(m, ref.reshape(frames, kobs, oh, ow, cout).mean(1)) <= 1.5 * e1 + 1e-7
    mean2 = torch.empty((frames, oh, ow, cout), device='cuda')
    C.conv_tile3_forward(mode, src.cuda(), ld, cin, frames, kobs, h, w, C.pack_conv_tile3_weights(mode, wk.cuda(), cin, cout, tn),
                         bias.cuda(), cout, tn, None, 0, mean2, cout, act=False, nprod=nprod)
    assert rel_l2(mean2.cpu(), pre64.reshape(frames, kobs, oh, ow, cout).mean(1)) <= 1.5 * e1 + 1e-7


@pytest.mark.parametrize('precision', ['f32x3', 'f32x3_9'])
@pytest.mark.parametrize('depth,uv,k,tn,mode', [(256, 128, 3, 64, 'test'), (1024, 256, 1, 64, 'test'), (256, 64, 2, 32, 'train')])
def test_model_with_three_term_split_encoder_vs_oracle(precision, depth, uv, k, tn, mode):
    """precision = f32x3 / f32x3_9 end to end: every eligible encoder conv on csrc/conv_tile3.hip, rendered texels against
    the fp32 oracle (bar 1e-6 -- the native path measures 1e-7) and against the native fp32 plan."""
    import nlt_amd
    from nlt_amd.models import get_model_class
    om, pm = make_pair(depth=depth, uv=uv, im=uv // 2, seed=depth + k + tn)
    p3 = get_model_class('nlt')(nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=uv // 2, imw=uv // 2, precision=precision))
    p3.load_weights(om.numpy_weights())
    p3.register_trainable()
    batch, nn = O.synth_batch(1, uv, uv, uv // 2, uv // 2, uv // 2, uv // 2, k=k, seed=30 + k)
    with torch.no_grad():
        o_vis = om.call(batch, mode, nn_list=nn)[3]
    nlev = sum(pm.net['query'].is_contracting) - 1
    hints = {'L%d.%s.%s' % (l, p, s): tn for l in range(1, nlev + 1) for p in 'qo' for s in ('s1', 's2')}
    vis = []
    for m in (pm, p3):
        m.plan.autotune = False
        m.plan.lds_hints = dict(hints)
        vis.append(m.call(to_device_batch(batch, nn), mode)[3])
    torch.cuda.synchronize()
    assert 'L3.o.s1' in p3.plan._ran_lds and 'L%d.q.s2' % nlev in p3.plan._ran_lds
    e_oracle, e_native = rel_l2(vis[1]['pred'].cpu(), o_vis['pred']), rel_l2(vis[1]['pred'].cpu(), vis[0]['pred'].cpu())
    print("%s depth %d: pred rel-L2 vs fp32 oracle %.2e, vs native fp32 plan %.2e (native vs oracle %.2e)"
          % (precision, depth, e_oracle, e_native, rel_l2(vis[0]['pred'].cpu(), o_vis['pred'])))
    assert e_oracle <= 1e-6


def test_conv_tile_rejects_what_it_cannot_do():
    x = torch.zeros(1, 8, 8, 24, device='cuda')
    assert not C.conv_tile_supported(C.CONV_K2S1, 24, 32, 32) and not C.conv_tile_supported(C.CONV_K2S1, 32, 48, 32)
    assert not C.conv_tile_supported(C.CONV1X1, 32, 32, 32) and not C.conv_tile_supported(C.CONV_K2S2, 32, 32, 16)
    with pytest.raises(C.NLTError):
        C.pack_conv_tile_weights(C.CONV_K2S1, torch.zeros(2, 2, 24, 32, device='cuda'), 24, 32, 32)
    packed = C.pack_conv_tile_weights(C.CONV_K2S2, torch.zeros(2, 2, 16, 32, device='cuda'), 16, 32, 32)
    with pytest.raises(C.NLTError):                      # odd input size for the stride-2 conv
        C.conv_tile_forward(C.CONV_K2S2, torch.zeros(1, 7, 8, 16, device='cuda'), 16, 16, 1, 1, 7, 8, packed,
                            torch.zeros(32, device='cuda'), 32, 32, torch.zeros(1, 3, 4, 32, device='cuda'), 32, None, 0)


@pytest.mark.parametrize('depth,uv,k,tn,mode', [(256, 64, 2, 32, 'test'), (256, 128, 3, 64, 'test'), (1024, 256, 1, 64, 'test'),
                                                 (256, 64, 2, 32, 'train')])
def test_model_with_lds_tiled_encoder_vs_oracle(depth, uv, k, tn, mode):
    om, pm = make_pair(depth=depth, uv=uv, im=uv // 2, seed=depth + k + tn)
    batch, nn = O.synth_batch(1, uv, uv, uv // 2, uv // 2, uv // 2, uv // 2, k=k, seed=30 + k)
    with torch.no_grad():
        o_vis = om.call(batch, mode, nn_list=nn)[3]
    pm.plan.autotune = False
    nlev = sum(pm.net['query'].is_contracting) - 1
    pm.plan.lds_hints = {'L%d.%s.%s' % (l, p, s): tn for l in range(1, nlev + 1) for p in 'qo' for s in ('s1', 's2')}
    p_vis = pm.call(to_device_batch(batch, nn), mode)[3]
    torch.cuda.synchronize()
    assert 'L3.o.s1' in pm.plan._ran_lds and 'L%d.q.s2' % nlev in pm.plan._ran_lds and ('L2.o.s1' in pm.plan._ran_lds) == (tn == 32)
    assert rel_l2(p_vis['pred'].cpu(), o_vis['pred']) <= 1e-4
