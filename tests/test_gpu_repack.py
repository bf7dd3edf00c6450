"""-m gpu: nlt_repack_weights (csrc/repack.hip) refills every packed buffer in one launch -- bit-identical to the per-layer
pack launches, including the adjoint-family fragments read straight from a SLICE of the forward layer's Keras array."""
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from nlt_amd.networks.elements import Conv2D, PackRegistry

pytestmark = pytest.mark.gpu


def test_one_launch_refresh_equals_the_per_layer_packs():
    rng = np.random.default_rng(0)
    R = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
    rows, expect = [], []

    def add(kind, mode, src, c0, c1, cout, tn=0, lo=0, full=None, ref=None):
        dst = torch.full_like(ref, float('nan'))
        rows.append(dict(src=src, dst=dst, kind=kind, mode=mode, c0=c0, c1=c1, cout=cout, tn=tn, lo=lo,
                         full=cout if full is None else full))
        expect.append((dst, ref))

    # forward-family fragments, single and dual source, every mode, ragged channel counts
    for mode, shape, c0, c1, cout in [(C.CONV1X1, (1, 1, 36, 3), 4, 32, 3), (C.CONV_K2S2, (2, 2, 32, 16), 32, 0, 16),
                                      (C.CONV_K2S1, (2, 2, 24, 40), 24, 0, 40), (C.DECONV_K2S2, (2, 2, 4, 40), 8, 32, 4),
                                      (C.DECONV_K2S1, (2, 2, 8, 8), 8, 0, 8), (C.DECONV_K2S2, (2, 2, 128, 1024), 512, 512, 128)]:
        w = R(*shape)
        add(C.REPACK_MFMA, mode, w, c0, c1, cout, ref=C.pack_conv_weights(mode, w, c0, c1, cout))
    # LDS-tile fragments
    for mode, cin, cout, tn in [(C.CONV_K2S1, 64, 64, 64), (C.CONV_K2S2, 16, 32, 32)]:
        w = R(2, 2, cin, cout)
        add(C.REPACK_TILE, mode, w, cin, 0, cout, tn=tn, ref=C.pack_conv_tile_weights(mode, w, cin, cout, tn))
    # adjoint fragments from slices of the forward array
    for fmode, shape, lo, hi in [(C.CONV_K2S2, (2, 2, 32, 16), 0, 32), (C.CONV_K2S1, (2, 2, 16, 16), 0, 16),
                                 (C.DECONV_K2S2, (2, 2, 8, 80), 0, 16), (C.DECONV_K2S2, (2, 2, 8, 80), 16, 80),
                                 (C.DECONV_K2S1, (2, 2, 64, 64), 0, 64), (C.CONV_K2S2, (2, 2, 512, 256), 128, 384)]:
        w = R(*shape)
        tr = fmode in (C.DECONV_K2S2, C.DECONV_K2S1)
        n_out = shape[2] if tr else shape[3]
        cin = shape[3] if tr else shape[2]
        ks = (w[..., lo:hi] if tr else w[:, :, lo:hi, :]).contiguous()
        adj = Conv2D.ADJOINT[fmode]
        add(C.REPACK_MFMA, adj, w, n_out, 0, hi - lo, lo=lo, full=cin, ref=C.pack_conv_weights(adj, ks, n_out, 0, hi - lo))
    table = C.repack_table(rows, 'cuda')
    C.repack_weights(*table)
    torch.cuda.synchronize()
    for i, (dst, ref) in enumerate(expect):
        assert torch.equal(dst, ref), i


def test_registry_refreshes_every_layout_of_a_layer_after_its_kernel_changes():
    conv = Conv2D(32, 2, 2)
    conv.build(32, 'cuda', seed=1)
    conv._registry = PackRegistry()
    a, t = conv.packed(32, 0), conv.packed_tile(32)
    adj, _ = conv.packed_adjoint(0, 32)
    ptrs = (a.data_ptr(), t.data_ptr(), adj.data_ptr())
    with torch.no_grad():
        conv.kernel.mul_(-2.0)                                            # an "optimizer step"
    a2 = conv.packed(32, 0)                                               # first stale access: ONE launch refreshes all three
    assert (a2.data_ptr(), conv.packed_tile(32).data_ptr(), conv.packed_adjoint(0, 32)[0].data_ptr()) == ptrs
    torch.cuda.synchronize()
    k = conv.kernel.detach()
    assert torch.equal(a2, C.pack_conv_weights(conv.mode, k, 32, 0, 32))
    assert torch.equal(conv.packed_tile(32), C.pack_conv_tile_weights(conv.mode, k, 32, 32, 32))
    assert torch.equal(conv.packed_adjoint(0, 32)[0], C.pack_conv_weights(C.DECONV_K2S2, k.contiguous(), 32, 0, 32))
