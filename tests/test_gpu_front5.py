"""-m gpu: fourth-generation front kernel (csrc/front5.hip: L1's stride-1 convs and level 2's stride-2 convs on the bf16 matrix
cores through the exact three-term split, persistent waves) against front4 (fp32 MFMA) on the same packed weights and inputs:
the results agree to the split's re-association (bound below: 1e-6 rel-L2, the parity bar of the f32x3_9 precision; measured
~1e-8), for float and uint8-store inputs, ragged sizes, missing neighbours, k = 1 (the item sequence alternates observation /
query) and inputs with more strips than resident waves (so that every wave walks several strips)."""
import pytest
import torch

from nlt_amd import capi as C
from gpu_util import make_pair

pytestmark = pytest.mark.gpu


def _weights(seed=0):
    _, pm = make_pair(depth=256, uv=64, im=32, seed=seed)
    pm.build('cuda')
    blob, blob_l2 = pm.plan._front_weights(torch.device('cuda'))
    assert blob_l2 is not None
    return pm, blob, blob_l2


def _outs(n, k, h, w):
    E = lambda *s: torch.full(s, float('nan'), device='cuda')
    return E(n, h // 2, w // 2, 32), E(n, h, w, 3), E(n, h // 4, w // 4, 32), E(n, k, h // 4, w // 4, 32)


def _close(ref, got, bound):
    for name, a, b in zip(('fm1', 'skip3', 'qtmp2', 'otmp2'), ref, got):
        assert not torch.isnan(b).any(), name
        rel = float((a.double() - b.double()).norm() / a.double().norm().clamp_min(1e-30))
        worst = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
        assert rel <= bound and worst <= 20 * bound, (name, rel, worst)


SHAPES = [(2, 1, 64, 96), (1, 2, 40, 72), (2, 4, 64, 64), (1, 3, 32, 32), (1, 4, 1024, 1024), (1, 4, 36, 100), (1, 6, 64, 64),
          (3, 1, 8, 8), (1, 2, 4, 4), (2, 1, 2048, 2048), (4, 2, 512, 768)]


@pytest.mark.parametrize('products,bound', [(9, 1e-6), (6, 1e-5)])
@pytest.mark.parametrize('n,k,h,w', SHAPES)
def test_front5_float_matches_front4(n, k, h, w, products, bound):
    pm, blob, blob_l2 = _weights(seed=k)
    g = torch.Generator(device='cuda').manual_seed(n * 1000 + k * 100 + h)
    U = lambda *s: torch.rand(s, device='cuda', generator=g)
    base, cvis, lvis = U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1)
    nn_rgb, nn_base = U(n, k, h, w, 3), U(n, k, h, w, 3)
    ref, got = _outs(n, k, h, w), _outs(n, k, h, w)
    C.front4_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, blob_l2, True, 0.3, *ref)
    C.front5_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, blob_l2, True, 0.3, *got, products)
    torch.cuda.synchronize()
    _close(ref, got, bound)


@pytest.mark.parametrize('n,k,h,w', [(2, 1, 64, 96), (3, 4, 64, 64), (1, 2, 40, 72), (2, 4, 512, 512), (1, 7, 32, 64), (4, 4, 1024, 1024)])
def test_front5_u8_store_is_bit_identical_to_front5_float_on_the_assembled_batch(n, k, h, w):
    """The uint8 -> float32 conversion is exact and the arithmetic behind it is the same instruction sequence."""
    pm, blob, blob_l2 = _weights(seed=10 + k)
    g = torch.Generator(device='cuda').manual_seed(7 * n + k + h)
    F = 6
    R = lambda *s: torch.randint(0, 256, s, device='cuda', generator=g, dtype=torch.uint8)
    diffuse, rgb, cvis, lvis = R(F, h, w, 3), R(F, h, w, 3), R(F, h, w), R(F, h, w)
    ids = torch.randint(0, F, (n,), device='cuda', generator=g, dtype=torch.int32)
    nn_ids = torch.randint(0, F, (n, k), device='cuda', generator=g, dtype=torch.int32)
    nn_ids[0, k - 1] = -1                                                     # a missing neighbour: zeros
    b = C.assemble_batch(diffuse, rgb, cvis, lvis, ids, nn_ids)
    ref4, ref, got = _outs(n, k, h, w), _outs(n, k, h, w), _outs(n, k, h, w)
    C.front4_forward(b['base'], b['cvis'], b['lvis'], b['nn_rgb'], b['nn_base'], n, k, h, w, blob, blob_l2, True, 0.3, *ref4)
    C.front5_forward(b['base'], b['cvis'], b['lvis'], b['nn_rgb'], b['nn_base'], n, k, h, w, blob, blob_l2, True, 0.3, *ref)
    C.front5_forward_u8(diffuse, rgb, cvis, lvis, ids, nn_ids, n, k, h, w, blob, blob_l2, True, 0.3, *got)
    torch.cuda.synchronize()
    _close(ref4, ref, 1e-6)
    for name, a, c in zip(('fm1', 'skip3', 'qtmp2', 'otmp2'), ref, got):
        assert not torch.isnan(c).any(), name
        assert torch.equal(a, c), (name, float((a - c).abs().max()))


def test_front5_is_deterministic_and_independent_of_what_the_buffers_held():
    pm, blob, blob_l2 = _weights(seed=3)
    n, k, h, w = 2, 4, 256, 512
    g = torch.Generator(device='cuda').manual_seed(11)
    U = lambda *s: torch.rand(s, device='cuda', generator=g)
    args = (U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1), U(n, k, h, w, 3), U(n, k, h, w, 3))
    a, b = _outs(n, k, h, w), _outs(n, k, h, w)
    C.front5_forward(*args, n, k, h, w, blob, blob_l2, False, 0.3, *a)
    for t in b:
        t.fill_(7.0)
    C.front5_forward(*args, n, k, h, w, blob, blob_l2, False, 0.3, *b)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_front5_rejects_what_it_cannot_take():
    pm, blob, blob_l2 = _weights()
    n, k, h, w = 1, 1, 32, 32
    Z = lambda *s: torch.zeros(s, device='cuda')
    outs = _outs(n, k, h, w)
    base = torch.zeros(n * h * w * 3 + 1, device='cuda')[1:].view(n, h, w, 3)          # 4-byte aligned only
    with pytest.raises(C.NLTError):
        C.front5_forward(base, Z(n, h, w, 1), Z(n, h, w, 1), Z(n, k, h, w, 3), Z(n, k, h, w, 3), n, k, h, w, blob, blob_l2,
                         True, 0.3, *outs)
    with pytest.raises(C.NLTError):                                                       # LeakyReLU slope outside [0, 1]
        C.front5_forward(Z(n, h, w, 3), Z(n, h, w, 1), Z(n, h, w, 1), Z(n, k, h, w, 3), Z(n, k, h, w, 3), n, k, h, w, blob,
                         blob_l2, True, 1.5, *outs)
    with pytest.raises(C.NLTError):                                                       # 7 products: not a form of the split
        C.front5_forward(Z(n, h, w, 3), Z(n, h, w, 1), Z(n, h, w, 1), Z(n, k, h, w, 3), Z(n, k, h, w, 3), n, k, h, w, blob,
                         blob_l2, True, 0.3, *outs, 7)
