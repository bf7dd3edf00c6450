"""-m gpu: second-generation front kernel (csrc/front4.hip) -- float inputs against front_kernel<true> (same folded
weights, same MFMA sequences; since r05 the biases are the accumulators' initial values instead of a separate add, so the
two kernels differ by that re-association: <= 5e-7 rel-L2, r04 review item 5), uint8-store inputs against the float kernel
on nlt_assemble_batch's output (<= 5e-7: byte-valued staging since r05), the train form against the inference form
(bit-identical), and the whole Model.call with the plan switch on / off."""
import numpy as np
import pytest
import torch

from nlt_amd import capi as C
from oracle import nlt_oracle as O
from gpu_util import make_pair, to_device_batch

pytestmark = pytest.mark.gpu
REASSOC = 5e-7          # bias-first vs bias-last accumulation of the same exact fp32 products


def close(a, b, tol=REASSOC):
    return float((a.double() - b.double()).norm()) <= tol * float(b.double().norm()) + 1e-30


def _weights(seed=0):
    _, pm = make_pair(depth=256, uv=64, im=32, seed=seed)
    pm.build('cuda')
    blob, blob_l2 = pm.plan._front_weights(torch.device('cuda'))
    assert blob_l2 is not None
    return pm, blob, blob_l2


def _outs(n, k, h, w):
    E = lambda *s: torch.full(s, float('nan'), device='cuda')
    return E(n, h // 2, w // 2, 32), E(n, h, w, 3), E(n, h // 4, w // 4, 32), E(n, k, h // 4, w // 4, 32)


@pytest.mark.parametrize('n,k,h,w', [(2, 1, 64, 96), (1, 2, 40, 72), (2, 4, 64, 64), (1, 3, 32, 32), (1, 4, 1024, 1024), (1, 4, 36, 100),
                                     (1, 6, 64, 64), (3, 1, 8, 8), (1, 2, 4, 4)])
def test_front4_float_matches_front2(n, k, h, w):
    """csrc/front4.hip (one wave per 4 x 16 strip, no workgroup barrier) against front_kernel<true>; k > 4 (which
    front_kernel<true> does not take) against the layer-by-layer plan instead."""
    pm, blob, blob_l2 = _weights(seed=k)
    g = torch.Generator(device='cuda').manual_seed(n * 1000 + k * 100 + h)
    U = lambda *s: torch.rand(s, device='cuda', generator=g)
    base, cvis, lvis = U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1)
    nn_rgb, nn_base = U(n, k, h, w, 3), U(n, k, h, w, 3)
    ref, got = _outs(n, k, h, w), _outs(n, k, h, w)
    C.front4_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, blob_l2, True, 0.3, *got)
    if k <= 4:
        C.front2_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, blob_l2, True, 0.3, *ref)
    else:                                   # observation by observation (k = 1 launches), means combined here
        fm_sum = torch.zeros_like(ref[0])
        for i in range(k):
            r1 = _outs(n, 1, h, w)
            C.front2_forward(base, cvis, lvis, nn_rgb[:, i:i + 1].contiguous(), nn_base[:, i:i + 1].contiguous(), n, 1, h, w,
                             blob, blob_l2, True, 0.3, *r1)
            ref[3][:, i] = r1[3][:, 0]
            fm_sum += r1[0]
        torch.cuda.synchronize()
        assert close(got[3], ref[3])
        assert not torch.isnan(got[0]).any() and not torch.isnan(got[1]).any() and not torch.isnan(got[2]).any()
        mean_ref = fm_sum[..., 16:] / k                                   # mean of the observations' level-1 maps
        assert float((mean_ref - got[0][..., 16:]).abs().max()) <= 1e-5 * float(mean_ref.abs().max())
        return
    torch.cuda.synchronize()
    for name, a, b in zip(('fm1', 'skip3', 'qtmp2', 'otmp2'), ref, got):
        assert not torch.isnan(b).any(), name
        assert close(b, a), (name, float((a - b).abs().max()))
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())          # ... and no single texel is off by more than a few ulps


@pytest.mark.parametrize('n,k,h,w', [(2, 1, 64, 96), (3, 4, 64, 64), (1, 2, 40, 72), (2, 4, 512, 512), (1, 7, 32, 64)])
def test_front4_u8_store_matches_float_on_assembled_batch(n, k, h, w):
    """r05: the uint8 variant feeds stage 1 the byte values and carries 1 / 255 in its weights (fl(W / 255) . u instead of
    W . fl(u / 255)): <= 5e-7 rel-L2 from the float kernel on nlt_assemble_batch's output (measured 1-3e-7), no texel off by more
    than a few ulps of the map's range; bit-identical until r04."""
    pm, blob, blob_l2 = _weights(seed=10 + k)
    g = torch.Generator(device='cuda').manual_seed(7 * n + k + h)
    F = 6
    R = lambda *s: torch.randint(0, 256, s, device='cuda', generator=g, dtype=torch.uint8)
    diffuse, rgb, cvis, lvis = R(F, h, w, 3), R(F, h, w, 3), R(F, h, w), R(F, h, w)
    ids = torch.randint(0, F, (n,), device='cuda', generator=g, dtype=torch.int32)
    nn_ids = torch.randint(0, F, (n, k), device='cuda', generator=g, dtype=torch.int32)
    nn_ids[0, k - 1] = -1                                                     # a missing neighbour: zeros
    b = C.assemble_batch(diffuse, rgb, cvis, lvis, ids, nn_ids)
    ref, got = _outs(n, k, h, w), _outs(n, k, h, w)
    C.front4_forward(b['base'], b['cvis'], b['lvis'], b['nn_rgb'], b['nn_base'], n, k, h, w, blob, blob_l2, True, 0.3, *ref)
    C.front4_forward_u8(diffuse, rgb, cvis, lvis, ids, nn_ids, n, k, h, w, blob, blob_l2, True, 0.3, *got)
    torch.cuda.synchronize()
    for name, a, c in zip(('fm1', 'skip3', 'qtmp2', 'otmp2'), ref, got):
        assert not torch.isnan(c).any(), name
        assert close(c, a), (name, float((a - c).abs().max()))
        assert float((a - c).abs().max()) <= 3e-6 * float(a.abs().max()), name


@pytest.mark.parametrize('n,k,h,w', [(2, 1, 64, 96), (1, 2, 40, 72), (1, 4, 64, 64), (1, 1, 1024, 1024), (1, 3, 36, 100)])
def test_front4_train_keeps_the_same_maps_as_the_first_generation_train_kernel(n, k, h, w):
    """nlt_front4_forward_train: fm1 / skip3 / qtmp2 / otmp2 bit-identical to nlt_front4_forward, and the three maps kept for
    the backward pass (obs1, qtmp1, otmp1) equal to what nlt_front_forward_train keeps up to the bias re-association (same
    folded weights, same MFMA sequences)."""
    pm, blob, blob_l2 = _weights(seed=k + 10)
    g = torch.Generator(device='cuda').manual_seed(n * 1000 + k * 100 + h)
    U = lambda *s: torch.rand(s, device='cuda', generator=g)
    base, cvis, lvis = U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1)
    nn_rgb, nn_base = U(n, k, h, w, 3), U(n, k, h, w, 3)
    E = lambda *s: torch.full(s, float('nan'), device='cuda')
    h2, w2 = h // 2, w // 2
    inf, tr = _outs(n, k, h, w), _outs(n, k, h, w)
    keep4 = (E(n, k, h2, w2, 16), E(n, h2, w2, 16), E(n, k, h2, w2, 16))                  # obs1, qtmp1, otmp1
    C.front4_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, blob_l2, True, 0.3, *inf)
    C.front4_forward_train(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, blob_l2, True, 0.3, *tr, *keep4)
    fm1, obs1, skip3, qtmp1, otmp1 = E(n, h2, w2, 32), E(n, k, h2, w2, 16), E(n, h, w, 3), E(n, h2, w2, 16), E(n, k, h2, w2, 16)
    ok_old = w % 8 == 0                                                                   # (the first generation's train form needs w / 2 in fours)
    if ok_old:
        C.front_forward_train(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, blob, True, 0.3, fm1, obs1, skip3, qtmp1, otmp1)
    torch.cuda.synchronize()
    for name, a, b in zip(('fm1', 'skip3', 'qtmp2', 'otmp2'), inf, tr):
        assert torch.equal(a, b), name
    assert not any(torch.isnan(t).any() for t in keep4)
    if ok_old:
        for name, a, b in zip(('obs1', 'qtmp1', 'otmp1'), (obs1, qtmp1, otmp1), keep4):
            assert close(b, a), (name, float((a - b).abs().max()))
        assert close(tr[0], fm1) and torch.equal(skip3, tr[1])                          # (skip3: VALU in both, no bias chain)


def test_front4_rejects_what_it_cannot_take():
    pm, blob, blob_l2 = _weights()
    n, k, h, w = 1, 1, 32, 32
    Z = lambda *s: torch.zeros(s, device='cuda')
    outs = _outs(n, k, h, w)
    base = torch.zeros(n * h * w * 3 + 1, device='cuda')[1:].view(n, h, w, 3)          # 4-byte aligned only
    assert not C.front4_supported(base)
    with pytest.raises(C.NLTError):
        C.front4_forward(base, Z(n, h, w, 1), Z(n, h, w, 1), Z(n, k, h, w, 3), Z(n, k, h, w, 3), n, k, h, w, blob, blob_l2,
                         True, 0.3, *outs)
    with pytest.raises(C.NLTError):                                                       # LeakyReLU slope outside [0, 1]
        C.front4_forward(Z(n, h, w, 3), Z(n, h, w, 1), Z(n, h, w, 1), Z(n, k, h, w, 3), Z(n, k, h, w, 3), n, k, h, w, blob,
                         blob_l2, True, 1.5, *outs)
    with pytest.raises(C.NLTError):                                                       # h not a multiple of 4
        C.front4_forward(Z(n, 30, w, 3), Z(n, 30, w, 1), Z(n, 30, w, 1), Z(n, k, 30, w, 3), Z(n, k, 30, w, 3), n, k, 30, w, blob,
                         blob_l2, True, 0.3, *outs)


def test_model_call_same_result_with_either_front_kernel():
    om, pm = make_pair(depth=256, uv=128, im=64, seed=5)
    batch, nn = O.synth_batch(2, 128, 128, 64, 64, 64, 64, k=4, seed=6)
    db = to_device_batch(batch, nn)
    res = []
    for v4 in (False, True):
        pm.plan.front_v4 = v4
        pm.plan._drop_tapes()
        out = pm.call(db, 'test')
        torch.cuda.synchronize()
        res.append((out[0].clone(), out[3]['pred'].clone()))
    assert close(res[1][0], res[0][0], 1e-6) and close(res[1][1], res[0][1], 1e-6)


def test_model_call_on_a_store_resident_batch_equals_the_eager_float_batch():
    """Dataset.load_batch(resident=True) -> Model.call: the front kernel reads the uint8 store itself; everything the
    call returns must equal, bit for bit, what the eager (assembled float32) batch gives -- in 'test' and 'vali' mode,
    and in 'train' mode (where the resident batch is materialised for the training kernels)."""
    import nlt_amd
    from nlt_amd.datasets import get_dataset_class
    from nlt_amd.datasets.synth import synthetic_store
    from nlt_amd.models import get_model_class
    uv, cam, k = 128, 64, 3
    store = synthetic_store(7, uv, cam, seed=3, k=k)
    cfg = nlt_amd.make_config(depth=256, uvh=uv, uvw=uv, imh=cam, imw=cam, bs=2)
    pm = get_model_class('nlt')(cfg).build('cuda')
    pm.register_trainable()
    ds = get_dataset_class('nlt')(cfg, 'train', store, k=k, ring=0)
    ids = store['ids'][2:4]
    eager, res = ds.load_batch(ids), ds.load_batch(ids, resident=True)
    assert res[2] is None and res[4] is None and eager[8].shape == (2, k, uv, uv, 3)
    for mode in ('test', 'vali'):
        a, b = pm.call(eager, mode, want_indices=True), pm.call(res, mode, want_indices=True)
        torch.cuda.synchronize()
        assert close(b[0], a[0], 1e-6) and close(b[3]['pred'], a[3]['pred'], 1e-6)     # (r05: byte-valued staging, 1 / 255 in the weights)
        # base and the uv2cam map gathered straight from the uint8 / fp16 stores (nlt_warp_forward_store): same bits,
        # same integer UV indices
        assert torch.equal(a[3]['base_camspc'], b[3]['base_camspc']) and torch.equal(a[3]['uv_indices'], b[3]['uv_indices'])
        if mode == 'vali':
            assert torch.equal(a[1], b[1]) and torch.equal(a[3]['gt'], b[3]['gt'])
    first = b[0].clone()
    for _ in range(3):                                          # recorded launch tape, then replays
        b = pm.call(res, 'test')
    torch.cuda.synchronize()
    assert torch.equal(first, b[0])
    grads = []
    for batch in (eager, res):
        pred, gt, kw, _ = pm(batch, mode='train')
        loss = pm.compute_loss(pred, gt, keep_batch=True).sum() / 2
        pm.flat_params.grad = None
        loss.backward()
        torch.cuda.synchronize()
        grads.append((float(loss.detach()), pm.flat_params.grad.clone()))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * abs(grads[0][0])          # (the loss reductions use float atomics)
    assert float((grads[0][1] - grads[1][1]).abs().max()) <= 1e-6 * float(grads[0][1].abs().max())   # (atomics in the warp adjoint)
