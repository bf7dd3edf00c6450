"""-m gpu: bf16 1x1 channel mix (csrc/chmix_bf16.hip, v_mfma_f32_16x16x32_bf16) against the oracle on bf16-rounded
operands.  Tolerance: the fp32 sums differ only by summation order (<= ~1e-5 absolute at these magnitudes), so outputs
agree except where that difference straddles a bf16 rounding boundary -- at most 1 bf16 ulp, on a small fraction of
elements."""
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def ulp_bf16(ref):
    a = ref.float().abs().clamp_min(2.0 ** -126)
    return torch.pow(2.0, torch.floor(torch.log2(a)) - 7)


@pytest.mark.parametrize('cin,cout,shape,act', [(64, 64, (1, 33, 47), True), (32, 32, (2, 16, 16), True), (128, 128, (1, 8, 24), False),
                                                 (64, 32, (1, 5, 3), True), (32, 128, (1, 1, 1), True), (64, 64, (2, 256, 256), True)])
def test_chmix_bf16_vs_oracle(cin, cout, shape, act):
    g = torch.Generator().manual_seed(cin + cout + shape[1])
    x = (torch.randn(shape + (cin,), generator=g) * 1.5).to(torch.bfloat16)
    w = torch.randn((1, 1, cin, cout), generator=g) * (cin ** -0.5)
    b = torch.randn(cout, generator=g) * 0.1
    ref = T.conv1x1_bf16(x, w, b, act)
    packed = C.chmix_bf16_pack(w.cuda())
    got = C.chmix_bf16_forward(x.cuda(), packed, b.cuda(), cout, act=act).cpu()
    assert got.shape == ref.shape and got.dtype == torch.bfloat16
    diff = (got.float() - ref.float()).abs()
    # 1 bf16 ulp of the result, plus the fp32 summation-order slack (<= cin * 2^-24 * sum|x w| ~ 1e-5 here), which is
    # what decides the few results that cancel to nearly zero
    assert bool((diff <= ulp_bf16(ref) * 1.001 + 1e-5).all()), float((diff / ulp_bf16(ref)).max())
    assert float((diff > 0).float().mean()) < 0.02                    # rounding-boundary cases only
    rel = float((got.float() - ref.float()).norm() / ref.float().norm())
    assert rel < 1e-3


def test_chmix_bf16_linearity_and_identity_at_2048():
    """BASELINE config 5 size (2048^2, 64 channels): identity weights copy the input exactly; scaling the weights
    by 2 doubles the (un-activated) output exactly (powers of two are exact in bf16)."""
    x = (torch.randn(1, 2048, 2048, 64, device='cuda') * 0.7).to(torch.bfloat16)
    eye = torch.eye(64, device='cuda').reshape(1, 1, 64, 64)
    z = torch.zeros(64, device='cuda')
    y = C.chmix_bf16_forward(x, C.chmix_bf16_pack(eye), z, 64, act=False)
    assert torch.equal(y, x)
    w = torch.randn(1, 1, 64, 64, device='cuda') * 0.125
    y1 = C.chmix_bf16_forward(x, C.chmix_bf16_pack(w), z, 64, act=False)
    y2 = C.chmix_bf16_forward(x, C.chmix_bf16_pack(2 * w), z, 64, act=False)
    assert torch.equal(y2.float(), 2 * y1.float())


def test_chmix_bf16_rejects_other_shapes():
    with pytest.raises(C.NLTError):
        C.chmix_bf16_pack(torch.zeros(1, 1, 48, 64, device='cuda'))
    with pytest.raises(C.NLTError):
        C.chmix_bf16_forward(torch.zeros(4, 64, device='cuda'), torch.zeros(4096, device='cuda', dtype=torch.bfloat16),
                             torch.zeros(64, device='cuda'), 64)
