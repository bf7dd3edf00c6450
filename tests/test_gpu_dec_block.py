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


# (c, 2c, 8c): the U-Net's own widths at these levels = the balanced kernel (dec_block10_kernel); everything else = the generic one
@pytest.mark.parametrize('c,cx,cs,n,h,w', [(8, 16, 64, 2, 32, 48), (16, 32, 128, 1, 24, 40), (8, 16, 64, 1, 9, 17), (16, 32, 128, 2, 8, 16),
                                           (8, 8, 32, 1, 20, 12), (16, 20, 36, 1, 11, 7), (8, 16, 64, 1, 256, 256),
                                           (8, 16, 32, 2, 32, 48), (16, 32, 64, 1, 24, 40), (8, 16, 32, 1, 9, 17), (16, 32, 64, 2, 8, 16),
                                           (8, 16, 32, 1, 21, 13), (16, 32, 64, 1, 11, 7), (8, 16, 32, 1, 256, 256), (16, 32, 64, 2, 128, 128),
                                           (16, 32, 64, 1, 1, 1), (8, 16, 32, 1, 3, 35), (16, 32, 128, 1, 1, 1), (8, 16, 64, 1, 3, 35), (16, 32, 128, 2, 128, 128),
                                           (8, 16, 64, 1, 21, 13), (16, 32, 128, 1, 11, 7)])
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


@pytest.mark.parametrize('c,n,h,w', [(8, 2, 64, 80), (16, 2, 40, 72), (8, 1, 9, 17), (16, 1, 13, 5)])
def test_balanced_kernel_is_bit_identical_to_the_generic_one(c, n, h, w, monkeypatch):
    """dec_block10_kernel keeps dec_block_kernel's accumulation order.  The generic kernel is reached through its own widths
    rule: the same tensors with the skip map split off differently cannot be used (the rule is on cx / cs), so the generic
    kernel is selected by NLT_DEC_GENERIC in a child interpreter instead."""
    import os, subprocess, sys, tempfile
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from nlt_amd import capi as C
from oracle import tf_ops as T
c, n, h, w = %d, %d, %d, %d
rng = np.random.default_rng(5)
d = lambda a: torch.from_numpy(a).cuda().contiguous()
x, skip = d(rng.standard_normal((n, h, w, 2 * c), dtype=np.float32)), d(rng.standard_normal((n, h, w, 8 * c), dtype=np.float32))
w2, w1 = d(T.glorot_uniform(rng, (2, 2, c, 10 * c))), d(T.glorot_uniform(rng, (2, 2, c, c)))
b2, b1 = d(rng.uniform(-0.1, 0.1, c).astype(np.float32)), d(rng.uniform(-0.1, 0.1, c).astype(np.float32))
out = torch.full((n, 2 * h, 2 * w, c), float('nan'), device='cuda')
C.dec_block_forward(x, 2 * c, skip, 8 * c, n, h, w, w2, b2, w1, b1, c, 0.3, out)
torch.cuda.synchronize()
np.save(sys.argv[1], out.cpu().numpy())
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), c, n, h, w)
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for generic in ('0', '1'):
            f = os.path.join(td, 'o%s.npy' % generic)
            subprocess.run([sys.executable, '-c', code, f], check=True, env=dict(os.environ, NLT_DEC_GENERIC=generic), timeout=300)
            outs.append(np.load(f))
    assert not np.isnan(outs[0]).any()
    assert np.array_equal(outs[0], outs[1])


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
