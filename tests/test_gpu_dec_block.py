"""-m gpu: fused expanding block (csrc/dec_block.hip: Conv2DTranspose k2s2 + LeakyReLU + Conv2DTranspose k2s1 + LeakyReLU,
intermediate in LDS) against the oracle's Keras restatement (oracle/tf_ops.py) and against the two-launch plan."""
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from oracle import tf_ops as T
from oracle import nlt_oracle as O
from gpu_util import rel_l2, make_pair, to_device_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('c,cx,cs,n,h,w', [(8, 16, 64, 2, 32, 48), (16, 32, 128, 1, 24, 40), (8, 16, 64, 1, 9, 17), (16, 32, 128, 2, 8, 16),
                                           (8, 8, 32, 1, 20, 12), (16, 20, 36, 1, 11, 7), (8, 16, 64, 1, 256, 256)])
def test_dec_block_matches_the_two_transposed_convs(c, cx, cs, n, h, w):
    rng = np.random.default_rng(c * 100 + h)
    x = torch.from_numpy(rng.standard_normal((n, h, w, cx), dtype=np.float32))
    skip = torch.from_numpy(rng.standard_normal((n, h, w, cs), dtype=np.float32))
    w2 = torch.from_numpy(T.glorot_uniform(rng, (2, 2, c, cx + cs)))
    w1 = torch.from_numpy(T.glorot_uniform(rng, (2, 2, c, c)))
    b2 = torch.from_numpy(rng.uniform(-0.1, 0.1, c).astype(np.float32))
    b1 = torch.from_numpy(rng.uniform(-0.1, 0.1, c).astype(np.float32))
    ref = T.leaky_relu(T.conv2d_transpose_same(T.leaky_relu(T.conv2d_transpose_same(torch.cat((x, skip), 3), w2, b2, 2)), w1, b1, 1))
    out = torch.full((n, 2 * h, 2 * w, c), float('nan'), device='cuda')
    d = lambda t: t.cuda().contiguous()
    C.dec_block_forward(d(x), cx, d(skip), cs, n, h, w, d(w2), d(b2), d(w1), d(b1), c, 0.3, out)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    assert rel_l2(out.cpu(), ref) <= 1e-5


def test_dec_block_rejects_other_widths():
    Z = lambda *s: torch.zeros(s, device='cuda')
    with pytest.raises(C.NLTError):
        C.dec_block_forward(Z(1, 8, 8, 8), 8, Z(1, 8, 8, 32), 32, 1, 8, 8, Z(2, 2, 4, 40), Z(4), Z(2, 2, 4, 4), Z(4), 4, 0.3, Z(1, 16, 16, 4))


def test_plan_uses_the_fused_blocks_and_matches_the_two_launch_plan():
    om, pm = make_pair(depth=256, uv=128, im=64, seed=3)
    batch, nn = O.synth_batch(2, 128, 128, 64, 64, 64, 64, k=2, seed=4)
    db = to_device_batch(batch, nn)
    with torch.no_grad():
        ref = om.call(batch, 'test', nn_list=nn)[3]['pred']
    from nlt_amd.engine import OpTimer
    res = {}
    for fused in (True, False):
        pm.plan.fuse_dec = fused
        pm.plan._drop_tapes()
        pm.call(db, 'test')
        pm.plan.timer = OpTimer()
        out = pm.call(db, 'test')
        torch.cuda.synchronize()
        labels = set(pm.plan.timer.collect())
        pm.plan.timer = None
        res[fused] = (out[3]['pred'].clone(), labels)
    assert 'L10.q' in res[True][1] and 'L11.q' in res[True][1] and 'L11.q.s1' not in res[True][1]
    assert 'L10.q.s2' in res[False][1] and 'L11.q.s1' in res[False][1] and 'L11.q' not in res[False][1]
    assert rel_l2(res[True][0].cpu(), ref) <= 1e-4 and rel_l2(res[True][0].cpu(), res[False][0].cpu()) <= 1e-5
