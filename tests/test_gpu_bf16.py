"""-m gpu: the bf16 middle of the network (BASELINE config 5; csrc/conv_bf16.hip): encoder levels >= 3 and the expanding
blocks mirroring them on v_mfma_f32_16x16x32_bf16 with bf16-stored activations, fp32 accumulation, fp32 ends.

Oracle: OracleModel.set_precision('bf16') rounds the SAME operands to bf16 at the same places (inputs as stored, weights
as packed, every stored map, the observation mean) and accumulates in fp32, so the HIP path differs from it only by fp32
summation order -- which can move a result across a bf16 rounding boundary (one bf16 ulp = 2^-8 relative) for a few
elements.  Stated tolerance: rel-L2 <= 5e-3 on the maps inside / leaving the bf16 region (measured 0.8e-3 on its output, 2.4e-3 on the bottleneck map after eight stacked bf16 layers), <= 1e-4 on the
rendered texels (where the region's contribution is one of several).  The distance to the fp32 oracle is reported."""
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
TOL_REGION, TOL_PRED = 5e-3, 1e-4


@pytest.mark.parametrize('mode,c0,c1,cout,n,h,w,f32_in,f32_out', [
    (C.CONV_K2S2, 64, 0, 64, 2, 16, 24, True, False), (C.CONV_K2S1, 64, 0, 64, 1, 9, 17, False, False),
    (C.CONV_K2S2, 32, 0, 64, 3, 8, 8, True, False), (C.DECONV_K2S2, 128, 256, 64, 1, 6, 10, False, False),
    (C.DECONV_K2S1, 16, 0, 16, 2, 12, 20, False, True), (C.DECONV_K2S2, 32, 128, 16, 1, 8, 12, False, False),
    (C.CONV_K2S1, 256, 0, 256, 1, 4, 4, False, False), (C.CONV1X1, 64, 0, 64, 1, 7, 5, False, True)])
def test_conv_bf16_matches_bf16_operand_oracle(mode, c0, c1, cout, n, h, w, f32_in, f32_out):
    rng = np.random.default_rng(mode * 1000 + c0 + h)
    rb = O.round_bf16
    x0 = torch.from_numpy(rng.standard_normal((n, h, w, c0), dtype=np.float32))
    x1 = torch.from_numpy(rng.standard_normal((n, h, w, c1), dtype=np.float32)) if c1 else None
    cin = c0 + c1
    tr = mode in (C.DECONV_K2S2, C.DECONV_K2S1)
    ks = 1 if mode == C.CONV1X1 else 2
    wk = torch.from_numpy(T.glorot_uniform(rng, (ks, ks, cout, cin) if tr else (ks, ks, cin, cout)))
    bias = torch.from_numpy(rng.uniform(-0.1, 0.1, cout).astype(np.float32))
    xin = rb(torch.cat((x0, x1), 3) if c1 else x0)
    stride = 2 if mode in (C.CONV_K2S2, C.DECONV_K2S2) else 1
    ref = (T.conv2d_transpose_same if tr else T.conv2d_same)(xin, rb(wk), bias, stride)
    ref = T.leaky_relu(ref, 0.3)
    if not f32_out:
        ref = rb(ref)
    dev = lambda t, lo: t.cuda().contiguous() if not lo else t.cuda().to(torch.bfloat16).contiguous()
    oh, ow = ref.shape[1:3]
    out = torch.zeros((n, oh, ow, cout), device='cuda', dtype=torch.float32 if f32_out else torch.bfloat16)
    packed = C.conv_bf16_pack(mode, wk.cuda(), c0, c1, cout)
    C.conv_bf16_forward(mode, dev(x0, not f32_in), c0, c0, dev(x1, True) if c1 else None, c1, c1, n, h, w, packed, bias.cuda(), cout,
                        out, cout, act=True, alpha=0.3)
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert rel_l2(got, ref) <= (2e-6 if f32_out else TOL_REGION)
    if not f32_out:                                           # at most one bf16 ulp apart, and only for a small fraction
        bad = (got != ref)
        assert float(bad.float().mean()) < 0.02
        assert float(((got - ref).abs() / ref.abs().clamp_min(1e-20))[bad].max() if bad.any() else 0.0) <= 2 ** -7


def test_obs_mean_bf16():
    n, k, hw, c = 2, 3, 35, 64
    x = torch.randn(n, k, hw, c).to(torch.bfloat16)
    fm = torch.zeros(n, hw, 2 * c, dtype=torch.bfloat16, device='cuda')
    C.obs_mean_bf16(x.cuda(), n, k, hw, c, fm.view(-1)[c:], 2 * c)
    torch.cuda.synchronize()
    ref = O.round_bf16(x.float().sum(1) * (1.0 / k))
    assert torch.equal(fm[..., c:].float().cpu(), ref) and not fm[..., :c].any()


@pytest.mark.parametrize('uv,k,n', [(128, 2, 2), (256, 4, 1)])
def test_bf16_model_vs_bf16_oracle_and_fp32_oracle(uv, k, n):
    om = O.OracleModel(depth=256, uvh=uv, uvw=uv, imh=uv // 2, imw=uv // 2, seed=3)
    pm = get_model_class('nlt')(nlt_amd.make_config(depth=256, uvh=uv, uvw=uv, imh=uv // 2, imw=uv // 2, precision='bf16'))
    pm.load_weights(om.numpy_weights())
    pm.register_trainable()
    # trained networks carry signal through the middle; random-init ones barely do: scale the region's kernels up so that
    # the comparison of the rendered texels actually sees it
    batch, nn = O.synth_batch(n, uv, uv, uv // 2, uv // 2, uv // 2, uv // 2, k=k, seed=8)
    with torch.no_grad():
        ref32 = om.call(batch, 'test', nn_list=nn)[3]['pred']
        om.set_precision('bf16')
        outs = []
        x = torch.cat((batch[1], batch[2], batch[3]), 3)
        om._call(x, [r - b for b, r in nn], layer_outputs=outs)
        ref16 = om.call(batch, 'test', nn_list=nn)[3]['pred']
    db = to_device_batch(batch, nn)
    for _ in range(3):                                        # autotune pass, tape record, tape replay
        got = pm.call(db, 'test')
    torch.cuda.synchronize()
    bufs = next(iter(pm.plan._bufs.values()))
    D = pm.plan.n_down
    region_out = bufs['dec'][D - 3]                           # output of the last bf16 block (16 channels, fp32)
    assert region_out.dtype == torch.float32 and bufs['fm'][3].dtype == torch.bfloat16 and bufs['fm'][2].dtype == torch.float32
    e_region = rel_l2(region_out.cpu(), outs[D + 1 + D - 3])
    e_fm = rel_l2(bufs['fm'][D].float().cpu()[..., :256], outs[D])
    e16, e32 = rel_l2(got[3]['pred'].cpu(), ref16), rel_l2(got[3]['pred'].cpu(), ref32)
    print("bf16 path: region output vs bf16 oracle %.2e, bottleneck map %.2e, pred vs bf16 oracle %.2e, pred vs fp32 oracle %.2e"
          % (e_region, e_fm, e16, e32))
    assert e_region <= TOL_REGION and e_fm <= TOL_REGION and e16 <= TOL_PRED
    with pytest.raises(C.NLTError):                           # training stays fp32
        pm.build('cuda')
        pm(db, mode='train')
