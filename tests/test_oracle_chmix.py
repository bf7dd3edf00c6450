"""CPU: the bf16 channel-mix oracle against an independent NumPy restatement (bit manipulation for the bf16 rounding)."""
import numpy as np
import torch

from oracle import tf_ops as T


def to_bf16_np(a):
    """float32 -> bf16 bit pattern -> float32, round to nearest even."""
    u = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def test_conv1x1_bf16_matches_numpy():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 5, 7, 32)).astype(np.float32)
    w = (rng.standard_normal((1, 1, 32, 64)) * 0.2).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32) * 0.1
    xb = torch.from_numpy(x).to(torch.bfloat16)
    assert np.array_equal(xb.float().numpy(), to_bf16_np(x))                       # torch's rounding == RNE
    ref = to_bf16_np(x).astype(np.float64) @ to_bf16_np(w[0, 0]).astype(np.float64) + b
    ref = np.where(ref > 0, ref, 0.3 * ref)
    got = T.conv1x1_bf16(xb, torch.from_numpy(w), torch.from_numpy(b)).float().numpy()
    assert np.array_equal(got, to_bf16_np(ref.astype(np.float32)))
    lin = T.conv1x1_bf16(xb, torch.from_numpy(w), torch.from_numpy(b), act=False).float().numpy()
    assert np.array_equal(lin, to_bf16_np((to_bf16_np(x).astype(np.float64) @ to_bf16_np(w[0, 0]).astype(np.float64) + b).astype(np.float32)))
