"""CPU: lane-level NumPy emulation of csrc/conv_wino.hip -- the Winograd F(2x2, 2x2) form of the stride-1 k2 convs
(Conv2D k2s1 'same' and its transpose): weight transform and packing order (csrc/pack_common.h nlt_wino_fragment), the
per-thread window of the input transform, the V / U LDS layouts and the bank pattern of every LDS access, the ds_read_b64
fragment offsets with their permuted K, v_mfma_f32_16x16x4_f32's operand / result layout, the lane-local output transform,
stage / observation sequencing, the epilogues (bias + LeakyReLU + running observation mean; backward-data mask / accumulate),
ragged image edges.  Compared with the oracle's Conv2D / Conv2DTranspose 'same' (oracle/tf_ops.py)."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T

K2S1, DECONV_K2S1 = 2, 4
BY, BX = 4, 16
V_SLOTS = 9 * BY * 2 * BX
LANE = np.arange(64); KK, J = LANE >> 4, LANE & 15


def mfma(a, b, acc):
    """v_mfma_f32_16x16x4_f32: A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15]; D row (lane >> 4) * 4 + reg, col lane & 15."""
    A = np.zeros((16, 4), np.float32); B = np.zeros((4, 16), np.float32)
    A[J, KK] = a; B[KK, J] = b
    D = A @ B
    for r in range(4):
        acc[:, r] += D[KK * 4 + r, J]
    return acc


def wino_fragment(wk, idx, cin, cout, tnt, full, lo, transposed):
    flat = wk.reshape(-1)
    e, i, q = idx & 3, (idx >> 2) & 15, (idx >> 6) & 1
    r = idx >> 7
    ct = r % tnt; r //= tnt
    ps = r % 9; r //= 9
    nc8 = cin >> 3
    c8 = r % nc8; g = r // nc8
    c = c8 * 8 + q * 4 + e
    o = (g * tnt + ct) * 16 + i
    xi, nu = ps // 3, ps % 3
    v = np.float32(0)
    for a in range(int(xi == 2), int(xi != 0) + 1):
        for b in range(int(nu == 2), int(nu != 0) + 1):
            t = (1 - a) * 2 + (1 - b) if transposed else a * 2 + b
            v = np.float32(v + (flat[(t * full + lo + o) * cin + c] if transposed else flat[(t * cin + c) * full + lo + o]))
    return v


def pack(wk, cin, cout, tnt, full=None, lo=0, transposed=False):
    full = cout if full is None else full
    return np.array([wino_fragment(wk, idx, cin, cout, tnt, full, lo, transposed) for idx in range(9 * cin * cout)], np.float32)


def xcd_tile(b, nblocks):
    return b if nblocks & 7 else (b & 7) * (nblocks >> 3) + (b >> 3)


def write_banks_ok(slots):
    """ds_write_b128: serviced 8 contiguous lanes at a time over 32 banks (128 bytes)."""
    for g0 in range(0, len(slots), 8):
        banks = np.concatenate([(s * 4 + np.arange(4)) % 32 for s in slots[g0:g0 + 8]])
        assert len(set(banks.tolist())) == 32, slots[g0:g0 + 8]


def read_b64_banks_ok(byte_offsets):
    """ds_read_b64: two 32-lane halves, 64 banks of 4 bytes."""
    for half in (byte_offsets[:32], byte_offsets[32:]):
        banks = np.concatenate([(o // 4 + np.arange(2)) % 64 for o in half])
        assert len(set(banks.tolist())) == 64


def conv_wino(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, tn, ldo, ldm=0, act=True, alpha=0.3, want_mean=False,
              mask_src=None, ld_mask=0, accumulate=False, out_init=None):
    TR = mode == DECONV_K2S1
    tnt = tn // 16
    RT, CT = 2, tnt // 2
    U_SLOTS = 9 * tnt * 2 * 16
    STAGE = V_SLOTS + U_SLOTS
    NU = (U_SLOTS + 127) // 128
    MEAN = (not TR) and tnt == 2 and (kobs > 1 or want_mean)
    tiles_y, tiles_x = (h + 2 * BY - 1) // (2 * BY), (w + 2 * BX - 1) // (2 * BX)
    nc8 = cin // 8
    S = src.reshape(-1)
    P4 = packed.reshape(-1, 4)
    out = np.full(frames * kobs * h * w * ldo, np.nan, np.float32) if out_init is None else out_init.reshape(-1).copy()
    mean_out = np.full(frames * h * w * ldm, np.nan, np.float32) if want_mean else None
    ntiles = frames * tiles_y * tiles_x
    total_stages = nc8 * kobs
    in_frame = h * w
    tid = np.arange(256)
    lane, wave = tid & 63, tid >> 6
    kk, j = lane >> 4, lane & 15
    wn, wm = wave & 1, wave >> 1
    frag = ((kk >> 1) * 16 + j) * 16 + (kk & 1) * 8                   # bytes
    checked = False
    for bx in range(ntiles):
        for g in range(cout // tn):
            tile = xcd_tile(bx, ntiles)
            tx0 = (tile % tiles_x) * 2 * BX; tile //= tiles_x
            ty0 = (tile % tiles_y) * 2 * BY
            f = tile // tiles_y
            lds = np.full((2 * STAGE, 4), np.nan, np.float32)
            bj, bq, brow = tid & 15, (tid >> 4) & 1, (tid >> 5) & 3
            wy0, wx0 = ty0 + 2 * brow - int(TR), tx0 + 2 * bj - int(TR)
            v_slot = (brow * 2 + bq) * BX + bj
            ut = tid - 128

            def stage_into(q, buf):
                i, c8 = q // nc8, q % nc8
                base = buf * STAGE
                vs = []
                for t in range(128):                                   # waves 0-1: one (block, quad) window each
                    d = np.zeros((3, 3, 4), np.float32)
                    for r in range(3):
                        for s in range(3):
                            gy, gx = wy0[t] + r, wx0[t] + s
                            if 0 <= gy < h and 0 <= gx < w:
                                a0 = ((f * kobs + i) * in_frame + gy * w + gx) * ld + c8 * 8 + bq[t] * 4
                                d[r, s] = S[a0:a0 + 4]
                    e = np.stack([d[0] - d[1], d[1], d[2] - d[1]])      # rows
                    v = np.stack([e[:, 0] - e[:, 1], e[:, 1], e[:, 2] - e[:, 1]], 1)    # columns -> v[xi][nu]
                    for x in range(3):
                        for n in range(3):
                            lds[base + (x * 3 + n) * (BY * 2 * BX) + v_slot[t]] = v[x, n]
                    vs.append(v_slot[t])
                for t in range(128, 256):                              # waves 2-3: the stage's weight slots, copied linearly
                    for n in range(NU):
                        s_ = ut[t] + 128 * n
                        if s_ < U_SLOTS:
                            lds[base + V_SLOTS + s_] = P4[(g * nc8 + c8) * U_SLOTS + s_]
                return vs

            acc = np.zeros((9, RT, CT, 256, 4), np.float32)
            mean = np.zeros((RT, CT, 4, 256, 4), np.float32)
            vs = stage_into(0, 0)
            if not checked:
                write_banks_ok(vs[:64]); write_banks_ok(vs[64:])          # every position plane is a multiple of 128 slots further
                write_banks_ok(list(ut[128:192])); write_banks_ok(list(ut[192:256]))
                read_b64_banks_ok(frag[:64])
                checked = True
            for q in range(total_stages):
                if q + 1 < total_stages:
                    nxt = (q + 1, (q + 1) & 1)
                Vb = (q & 1) * STAGE * 16
                Ub = Vb + V_SLOTS * 16
                flat = lds.reshape(-1)
                for ps in range(9):
                    for wv in range(4):
                        sl = slice(wv * 64, wv * 64 + 64)
                        bf = [[flat[(Vb + ((ps * BY + wm[sl][0] * RT + rt) * 2 * BX) * 16 + frag[sl]) // 4 + s] for s in range(2)]
                              for rt in range(RT)]
                        af = [[flat[(Ub + ((ps * tnt + wn[sl][0] * CT + ct) * 2 * 16) * 16 + frag[sl]) // 4 + s] for s in range(2)]
                              for ct in range(CT)]
                        for s in range(2):
                            for rt in range(RT):
                                for ct in range(CT):
                                    acc[ps, rt, ct, sl] = mfma(af[ct][s], bf[rt][s], acc[ps, rt, ct, sl])
                if (q + 1) % nc8 == 0:
                    i = q // nc8
                    for ct in range(CT):
                        for rt in range(RT):
                            a_ = acc[:, rt, ct]
                            r0 = [a_[n] + a_[3 + n] for n in range(3)]
                            r1 = [a_[3 + n] + a_[6 + n] for n in range(3)]
                            y = [r0[0] + r0[1], r0[1] + r0[2], r1[0] + r1[1], r1[1] + r1[2]]
                            acc[:, rt, ct] = 0
                            for t in range(256):
                                oc = (g * tnt + wn[t] * CT + ct) * 16 + 4 * kk[t]
                                bv = bias[oc:oc + 4] if bias is not None else np.zeros(4, np.float32)
                                for uv in range(4):
                                    gy, gx = ty0 + 2 * (wm[t] * RT + rt) + (uv >> 1), tx0 + 2 * j[t] + (uv & 1)
                                    inside = gy < h and gx < w
                                    v = y[uv][t] + bv
                                    ot = ((f * kobs + i) * h + gy) * w + gx
                                    if mask_src is not None or accumulate:
                                        if inside:
                                            if accumulate:
                                                v = v + out[ot * ldo + oc: ot * ldo + oc + 4]
                                            if mask_src is not None:
                                                mk = mask_src.reshape(-1)[ot * ld_mask + oc: ot * ld_mask + oc + 4]
                                                v = v * np.where(mk > 0, np.float32(1), np.float32(alpha))
                                            out[ot * ldo + oc: ot * ldo + oc + 4] = v
                                        continue
                                    if act:
                                        v = np.where(v > 0, v, np.float32(alpha) * v)
                                    if MEAN:
                                        mean[rt, ct, uv, t] += v
                                    if inside:
                                        out[ot * ldo + oc: ot * ldo + oc + 4] = v
                                        if MEAN and want_mean and i == kobs - 1:
                                            mt = (f * h + gy) * w + gx
                                            mean_out[mt * ldm + oc: mt * ldm + oc + 4] = mean[rt, ct, uv, t] * np.float32(1.0 / kobs)
                if q + 1 < total_stages:
                    stage_into(*nxt)
    return out, mean_out


def _conv_ref(x, wk, b, alpha, transposed):
    xt, wt, bt = torch.tensor(x), torch.tensor(wk), torch.tensor(b)
    y = T.conv2d_transpose_same(xt, wt, bt, 1) if transposed else T.conv2d_same(xt, wt, bt, 1)
    return T.leaky_relu(y, alpha).numpy() if alpha is not None else y.numpy()


@pytest.mark.parametrize('tn,cin,cout,h,w,kobs', [(32, 8, 32, 10, 36, 1), (64, 16, 64, 8, 32, 1), (32, 16, 32, 7, 19, 3), (64, 8, 64, 9, 33, 1)])
def test_forward_conv_k2s1_matches_the_oracle(tn, cin, cout, h, w, kobs):
    rng = np.random.default_rng(tn + h)
    frames = 2
    src = rng.standard_normal((frames * kobs, h, w, cin)).astype(np.float32)
    wk = (rng.standard_normal((2, 2, cin, cout)) * 0.3).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32) * 0.1
    packed = pack(wk, cin, cout, tn // 16)
    want_mean = kobs > 1
    out, mean = conv_wino(K2S1, src, cin, cin, frames, kobs, h, w, packed, bias, cout, tn, cout, cout, True, 0.3, want_mean)
    ref = _conv_ref(src, wk, bias, 0.3, False)
    got = out.reshape(frames * kobs, h, w, cout)
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, ref, atol=2e-5 * np.abs(ref).max())
    if want_mean:
        np.testing.assert_allclose(mean.reshape(frames, h, w, cout), ref.reshape(frames, kobs, h, w, cout).mean(1),
                                   atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize('tn,cin,cout,h,w', [(32, 8, 32, 9, 33), (64, 16, 64, 8, 35)])
def test_forward_conv2dtranspose_k2s1_matches_the_oracle(tn, cin, cout, h, w):
    """Forward Conv2DTranspose k2s1 'same': (kh,kw,Cout,Cin) array, flipped taps, window one texel up / left."""
    rng = np.random.default_rng(tn + w)
    src = rng.standard_normal((2, h, w, cin)).astype(np.float32)
    wk = (rng.standard_normal((2, 2, cout, cin)) * 0.3).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32) * 0.1
    packed = pack(wk, cin, cout, tn // 16, transposed=True)
    out, _ = conv_wino(DECONV_K2S1, src, cin, cin, 2, 1, h, w, packed, bias, cout, tn, cout, 0, True, 0.3)
    ref = _conv_ref(src, wk, bias, 0.3, True)
    got = out.reshape(2, h, w, cout)
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, ref, atol=2e-5 * np.abs(ref).max())


def test_backward_data_of_a_conv_k2s1_slice_with_mask_and_accumulate():
    """Backward-data of a forward Conv2D k2s1 (Cin_f -> Cout_f) w.r.t. its input channels [lo, hi): the transposed family
    reading the conv's own (kh,kw,Cin_f,Cout_f) array as (kh,kw,N_full,K) with K = Cout_f, columns = the slice; epilogue:
    accumulate into the target, then the producer's LeakyReLU' mask."""
    rng = np.random.default_rng(5)
    cin_f, cout_f, lo, hi, h, w = 48, 16, 8, 40, 7, 34
    dpre = rng.standard_normal((2, h, w, cout_f)).astype(np.float32)
    wk = (rng.standard_normal((2, 2, cin_f, cout_f)) * 0.3).astype(np.float32)
    packed = pack(wk, cout_f, hi - lo, 2, full=cin_f, lo=lo, transposed=True)
    mask = rng.standard_normal((2, h, w, hi - lo)).astype(np.float32)
    prev = rng.standard_normal((2, h, w, hi - lo)).astype(np.float32)
    out, _ = conv_wino(DECONV_K2S1, dpre, cout_f, cout_f, 2, 1, h, w, packed, None, hi - lo, 32, hi - lo, act=False, alpha=0.3,
                       mask_src=mask, ld_mask=hi - lo, accumulate=True, out_init=prev)
    x = torch.zeros((2, h, w, cin_f), requires_grad=True)
    y = T.conv2d_same(x, torch.tensor(wk), torch.zeros(cout_f), 1)
    (gx,) = torch.autograd.grad(y, x, torch.tensor(dpre))
    ref = (prev + gx.numpy()[..., lo:hi]) * np.where(mask > 0, 1.0, 0.3).astype(np.float32)
    np.testing.assert_allclose(out.reshape(2, h, w, hi - lo), ref, atol=2e-5 * np.abs(ref).max())
