"""CPU: lane-level NumPy mirror of csrc/wgrad.hip's MFMA kernel (K tiles over tap x source
segments, N tiles incl. the DECONV_K2S2 (a,b,o) columns, row slices, Keras-layout write-back,
bias column sums) under the documented v_mfma_f32_16x16x4_f32 fragment layout."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T
from test_mfma_index_math import TAPS, keras_widx, chunks16, tap_texel, mfma, CONV1X1, CONV_K2S2, CONV_K2S1, DECONV_K2S2, DECONV_K2S1


def emulate_wgrad(mode, src0, ld0, c0, src1, ld1, c1, n, h, w, dp, ldp, cout, KT, NT, rows_per_split):
    gh, gw, oh, ow, N = h, w, h, w, cout
    if mode == CONV_K2S2:
        gh = oh = h // 2; gw = ow = w // 2
    if mode == DECONV_K2S2:
        oh, ow, N = 2 * h, 2 * w, 4 * cout
    M = n * gh * gw
    cin = c0 + c1
    ch0, ch1 = chunks16(c0), chunks16(c1)
    cps = ch0 + ch1
    ktiles, ntiles = TAPS[mode] * cps, (N + 15) >> 4
    kgroups, ngroups = -(-ktiles // KT), -(-ntiles // NT)
    msplits = -(-M // rows_per_split)
    dw = np.zeros(TAPS[mode] * cin * N, np.float64)
    db = np.zeros(cout, np.float64)
    for wave in range(kgroups * ngroups * msplits):
        ms = wave % msplits; rest = wave // msplits
        ng, kg = rest % ngroups, rest // ngroups
        acc = np.zeros((KT, NT, 64, 4), np.float32)
        bsum = np.zeros((NT, 64), np.float32)
        m_begin = ms * rows_per_split
        m_end = min(m_begin + rows_per_split, M)
        for m0 in range(m_begin, m_end, 4):
            a = np.zeros((KT, 64), np.float32); b = np.zeros((NT, 64), np.float32)
            for lane in range(64):
                li, mm = lane & 15, lane >> 4
                m = m0 + mm
                rv = m < m_end
                mc = m if rv else m_begin
                x, y, f = mc % gw, (mc // gw) % gh, mc // (gw * gh)
                for kt in range(KT):
                    tile = kg * KT + kt
                    if tile >= ktiles:
                        continue
                    t, r = tile // cps, tile % cps
                    s1 = r >= ch0
                    c = ((r - ch0) if s1 else r) * 16 + li
                    cs = c1 if s1 else c0
                    tex = tap_texel(mode, h, w, f, y, x, t)
                    if rv and c < cs and tex >= 0:
                        a[kt, lane] = (src1[tex * ld1 + c] if s1 else src0[tex * ld0 + c])
                for nt in range(NT):
                    col = (ng * NT + nt) * 16 + li
                    if col >= N or not rv:
                        continue
                    ab, oc = (col // cout, col % cout) if mode == DECONV_K2S2 else (0, col)
                    otex = mc if mode != DECONV_K2S2 else (f * oh + 2 * y + (ab >> 1)) * ow + 2 * x + (ab & 1)
                    b[nt, lane] = dp[otex * ldp + oc]
                    bsum[nt, lane] += b[nt, lane]
            for kt in range(KT):
                for nt in range(NT):
                    mfma(a[kt], b[nt], acc[kt, nt])
        for kt in range(KT):
            tile = kg * KT + kt
            if tile >= ktiles:
                continue
            t, r0 = tile // cps, tile % cps
            s1 = r0 >= ch0
            cs = c1 if s1 else c0
            for nt in range(NT):
                for lane in range(64):
                    li, mm = lane & 15, lane >> 4
                    col = (ng * NT + nt) * 16 + li
                    if col >= N:
                        continue
                    for r in range(4):
                        cl = ((r0 - ch0) if s1 else r0) * 16 + mm * 4 + r
                        if cl < cs:
                            dw[keras_widx(mode, t, (c0 if s1 else 0) + cl, col, cin, cout)] += acc[kt, nt, lane, r]
        if kg == 0:
            for nt in range(NT):
                for li in range(16):
                    col = (ng * NT + nt) * 16 + li
                    if col < N:
                        oc = col % cout if mode == DECONV_K2S2 else col
                        db[oc] += bsum[nt, li] + bsum[nt, li + 16] + bsum[nt, li + 32] + bsum[nt, li + 48]
    return dw, db


@pytest.mark.parametrize('mode,n,h,w,c0,c1,cout,KT,NT,rows', [
    (CONV1X1, 1, 3, 5, 16, 0, 16, 1, 1, 8),
    (CONV_K2S2, 1, 4, 6, 8, 4, 32, 4, 2, 4),
    (CONV_K2S1, 2, 3, 3, 16, 0, 16, 4, 1, 12),
    (DECONV_K2S2, 1, 2, 3, 8, 32, 4, 2, 1, 4),
    (DECONV_K2S2, 1, 3, 2, 16, 0, 16, 1, 4, 8),
    (DECONV_K2S1, 1, 3, 4, 4, 0, 4, 4, 1, 64),
])
def test_wgrad_kernel_index_math(mode, n, h, w, c0, c1, cout, KT, NT, rows):
    rng = np.random.default_rng(mode * 5 + cout)
    tr = mode in (DECONV_K2S2, DECONV_K2S1)
    k = 1 if mode == CONV1X1 else 2
    s = 2 if mode in (CONV_K2S2, DECONV_K2S2) else 1
    cin = c0 + c1
    pad0, pad1, padp = 4, 8, 4
    x0 = rng.standard_normal((n, h, w, c0 + pad0)).astype(np.float32)
    x1 = rng.standard_normal((n, h, w, c1 + pad1)).astype(np.float32)
    x = np.concatenate((x0[..., :c0], x1[..., :c1]), -1) if c1 else x0[..., :c0]
    wshape = (k, k, cout, cin) if tr else (k, k, cin, cout)
    wz = torch.zeros(wshape, requires_grad=True); bz = torch.zeros(cout, requires_grad=True)
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    y = f(torch.tensor(x), wz, bz, s)
    dp = rng.standard_normal(tuple(y.shape[:3]) + (cout + padp,)).astype(np.float32)
    gw, gb = torch.autograd.grad(y, (wz, bz), torch.tensor(dp[..., :cout]))
    dw, db = emulate_wgrad(mode, x0.reshape(-1), c0 + pad0, c0, x1.reshape(-1), c1 + pad1, c1, n, h, w,
                           dp.reshape(-1), cout + padp, cout, KT, NT, rows)
    np.testing.assert_allclose(dw.reshape(wshape), gw.numpy(), atol=1e-4)
    np.testing.assert_allclose(db, gb.numpy(), atol=1e-4)


# ---------------------------------------------------------------------------------------------------------
# second generation: csrc/wgrad_tile.hip (quad-permuted operands, workspace slices, deterministic reduce)
# ---------------------------------------------------------------------------------------------------------
def emulate_wgrad_tiled(mode, src0, ld0, c0, src1, ld1, c1, n, h, w, dp, ldp, cout, rows_per_split):
    gh, gw, oh, ow, N = h, w, h, w, cout
    if mode == CONV_K2S2:
        gh = oh = h // 2; gw = ow = w // 2
    if mode == DECONV_K2S2:
        oh, ow, N = 2 * h, 2 * w, 4 * cout
    M = n * gh * gw
    cin = c0 + c1
    q0, qpt = c0 >> 2, cin >> 2
    KQ, NQ = TAPS[mode] * qpt, N >> 2
    kblocks, nblocks = -(-KQ // 16), -(-NQ // 16)
    msplits = -(-M // rows_per_split)
    per_slice = kblocks * nblocks * 4096
    ws = np.zeros(msplits * per_slice, np.float32)
    wsb = np.zeros(msplits * nblocks * 64, np.float32)
    lane = np.arange(64); I, KKl = lane & 15, lane >> 4
    for wave in range(kblocks * nblocks * msplits):
        ms = wave % msplits; rest = wave // msplits
        nb, kb = rest % nblocks, rest // nblocks
        kq = kb * 16 + I
        a_ok = kq < KQ
        tap = np.where(a_ok, kq // qpt, 0); cq = np.where(a_ok, kq - tap * qpt, 0)
        from1 = cq >= q0
        nq = nb * 16 + I
        b_ok = nq < NQ
        ncol = np.where(b_ok, 4 * nq, 0)
        ab = ncol // cout if mode == DECONV_K2S2 else np.zeros(64, int)
        oc = ncol - ab * cout if mode == DECONV_K2S2 else ncol
        acc = np.zeros((4, 4, 64, 4), np.float32)
        bsum = np.zeros((64, 4), np.float32)
        m_begin = ms * rows_per_split; m_end = min(m_begin + rows_per_split, M)
        for m0 in range(m_begin, m_end, 4):
            av = np.zeros((64, 4), np.float32); bv = np.zeros((64, 4), np.float32)
            for l in range(64):
                m = m0 + KKl[l]
                rv = m < m_end
                mc = m if rv else m_begin
                x, y, f = mc % gw, (mc // gw) % gh, mc // (gw * gh)
                tex = tap_texel(mode, h, w, f, y, x, int(tap[l]))
                if rv and a_ok[l] and tex >= 0:
                    base = (src1, tex * ld1 + 4 * (cq[l] - q0)) if from1[l] else (src0, tex * ld0 + 4 * cq[l])
                    av[l] = base[0][base[1]:base[1] + 4]
                otex = mc if mode != DECONV_K2S2 else (f * oh + 2 * y + (ab[l] >> 1)) * ow + 2 * x + (ab[l] & 1)
                if rv and b_ok[l]:
                    bv[l] = dp[otex * ldp + oc[l]: otex * ldp + oc[l] + 4]
            bsum += bv
            for e in range(4):
                for f4 in range(4):
                    mfma(av[:, e], bv[:, f4], acc[e, f4])
        base = ((ms * kblocks + kb) * nblocks + nb) * 16
        for e in range(4):
            for f4 in range(4):
                for l in range(64):
                    a0 = ((base + e * 4 + f4) * 64 + l) * 4
                    ws[a0:a0 + 4] = acc[e, f4, l]
        if kb == 0:
            for i in range(16):
                tot = bsum[i] + bsum[i + 16] + bsum[i + 32] + bsum[i + 48]
                a0 = ((ms * nblocks + nb) * 16 + i) * 4
                wsb[a0:a0 + 4] = tot
    dw = np.zeros(TAPS[mode] * cin * N, np.float64)
    for idx in range(per_slice):
        r, l, ef = idx & 3, (idx >> 2) & 63, (idx >> 8) & 15
        blk = idx >> 12
        nb, kb = blk % nblocks, blk // nblocks
        kq = kb * 16 + 4 * (l >> 4) + r; nq = nb * 16 + (l & 15)
        if kq >= KQ or nq >= NQ:
            continue
        s = sum(ws[ms * per_slice + idx] for ms in range(msplits))
        tap = kq // qpt
        c = 4 * (kq - tap * qpt) + (ef >> 2)
        dw[keras_widx(mode, tap, c, 4 * nq + (ef & 3), cin, cout)] += s
    db = np.zeros(cout, np.float64)
    nab = 4 if mode == DECONV_K2S2 else 1
    for ocq in range(cout // 4):                                     # wgrad_bias_reduce_kernel: one workgroup per output quad
        for it in range(nab * msplits):
            ab, ms = it % nab, it // nab
            nq = ((ab * cout) >> 2) + ocq
            a0 = ((ms * nblocks + nq // 16) * 16 + nq % 16) * 4
            db[4 * ocq: 4 * ocq + 4] += wsb[a0:a0 + 4]
    return dw, db


@pytest.mark.parametrize('mode,n,h,w,c0,c1,cout,rows', [
    (CONV1X1, 1, 3, 5, 16, 0, 16, 8),
    (CONV_K2S2, 1, 4, 6, 8, 4, 32, 4),
    (CONV_K2S1, 2, 3, 3, 16, 0, 16, 12),
    (DECONV_K2S2, 1, 2, 3, 8, 32, 4, 4),
    (DECONV_K2S2, 1, 3, 2, 16, 0, 20, 8),           # 80 columns -> 20 n-quads: two n-blocks, the second ragged
    (DECONV_K2S1, 1, 3, 4, 4, 0, 4, 64),
    (CONV_K2S1, 1, 2, 3, 72, 0, 8, 4),              # 4 taps x 18 quads = 72 k-quads: 5 k-blocks
])
def test_wgrad_tiled_kernel_index_math(mode, n, h, w, c0, c1, cout, rows):
    rng = np.random.default_rng(mode * 7 + cout)
    tr = mode in (DECONV_K2S2, DECONV_K2S1)
    k = 1 if mode == CONV1X1 else 2
    s = 2 if mode in (CONV_K2S2, DECONV_K2S2) else 1
    cin = c0 + c1
    pad0, pad1, padp = 4, 8, 4
    x0 = rng.standard_normal((n, h, w, c0 + pad0)).astype(np.float32)
    x1 = rng.standard_normal((n, h, w, c1 + pad1)).astype(np.float32)
    x = np.concatenate((x0[..., :c0], x1[..., :c1]), -1) if c1 else x0[..., :c0]
    wshape = (k, k, cout, cin) if tr else (k, k, cin, cout)
    wz = torch.zeros(wshape, requires_grad=True); bz = torch.zeros(cout, requires_grad=True)
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    y = f(torch.tensor(x), wz, bz, s)
    dp = rng.standard_normal(tuple(y.shape[:3]) + (cout + padp,)).astype(np.float32)
    gw, gb = torch.autograd.grad(y, (wz, bz), torch.tensor(dp[..., :cout]))
    dw, db = emulate_wgrad_tiled(mode, x0.reshape(-1), c0 + pad0, c0, x1.reshape(-1), c1 + pad1, c1, n, h, w,
                                 dp.reshape(-1), cout + padp, cout, rows)
    np.testing.assert_allclose(dw.reshape(wshape), gw.numpy(), atol=1e-4)
    np.testing.assert_allclose(db, gb.numpy(), atol=1e-4)


# ---------------------------------------------------------------------------------------------------------
# narrow layers: csrc/wgrad_narrow.hip (MFMA tile = [K index 16] x [output channel 16]; scalar and quad-A operand forms,
# the wave-strided row walk, 4 waves -> one block per row slice -> workspace -> fixed-order reduction)
# ---------------------------------------------------------------------------------------------------------
def emulate_wgrad_narrow(mode, src0, ld0, c0, src1, ld1, c1, n, h, w, dp, ldp, cout, rows_per_split, quad):
    gh, gw, oh, ow, N = h, w, h, w, cout
    S = 1
    if mode == CONV_K2S2:
        gh = oh = h // 2; gw = ow = w // 2; S = 2
    if mode == DECONV_K2S2:
        oh, ow, N = 2 * h, 2 * w, 4 * cout
    M = n * gh * gw
    cin = c0 + c1
    taps = 1 if mode == DECONV_K2S2 else 4
    K = taps * cin
    NT = 1 if N <= 16 else 2
    if quad:
        MQ = 1 if K <= 64 else 2
        MT = 4 * MQ
    else:
        MT = 2 if K <= 32 else (4 if K <= 64 else 8)
    NC, KN = NT * 16, MT * 16 * NT * 16
    PER = KN + NC
    msplits = -(-M // rows_per_split)
    ws = np.zeros((msplits + 1) * PER, np.float32)
    lane = np.arange(64); I, KKl = lane & 15, lane >> 4
    sign = -1 if mode == DECONV_K2S1 else 1

    def a_value(kidx, m_row):                                        # X[row][k index] with the kernel's validity rules
        if kidx >= K:
            return 0.0
        tap, c = kidx // cin, kidx % cin
        a, b = (tap >> 1, tap & 1) if taps == 4 else (0, 0)
        x, y, f = m_row % gw, (m_row // gw) % gh, m_row // (gw * gh)
        iy, ix = S * y + sign * a, S * x + sign * b
        if not (0 <= iy < h and 0 <= ix < w):
            return 0.0
        tex = (f * h + iy) * w + ix
        return src1[tex * ld1 + c - c0] if c >= c0 else src0[tex * ld0 + c]

    for ms in range(msplits):
        m_begin = ms * rows_per_split
        m_end = min(m_begin + rows_per_split, M)
        block = np.zeros(PER, np.float32)
        for wv in range(4):
            acc = np.zeros((MT, NT, 64, 4), np.float32)              # [row tile][col tile][lane][r]
            bsum = np.zeros((NT, 64), np.float32)
            first = m_begin + 4 * wv
            nsteps = (m_end - first + 15) // 16 if first < m_end else 0
            for s in range(nsteps):
                m = first + 16 * s + KKl                             # the wave's steps are 16 rows apart
                rv = m < m_end
                av = np.zeros((MT, 64), np.float32); bv = np.zeros((NT, 64), np.float32)
                for l in range(64):
                    if not rv[l]:
                        continue
                    for mt in range(MT):
                        if quad:                                     # tile e of quad block mq holds K index 4 * (16 mq + i) + e
                            mq, e = mt >> 2, mt & 3
                            kidx = 4 * (16 * mq + I[l]) + e
                        else:
                            kidx = 16 * mt + I[l]
                        av[mt, l] = a_value(kidx, m[l])
                    x, y, f = m[l] % gw, (m[l] // gw) % gh, m[l] // (gw * gh)
                    for nt in range(NT):
                        col = nt * 16 + I[l]
                        if col >= N:
                            continue
                        ab, oc = (col // cout, col % cout) if mode == DECONV_K2S2 else (0, col)
                        otex = m[l] if mode != DECONV_K2S2 else (f * oh + 2 * y + (ab >> 1)) * ow + 2 * x + (ab & 1)
                        bv[nt, l] = dp[otex * ldp + oc]
                for nt in range(NT):
                    bsum[nt] += bv[nt]
                    for mt in range(MT):
                        mfma(av[mt], bv[nt], acc[mt, nt])
            for mt in range(MT):                                     # accumulator (mt, D row 4 kk + r) -> K index
                for nt in range(NT):
                    for l in range(64):
                        for r in range(4):
                            row = 4 * KKl[l] + r
                            kidx = 4 * (16 * (mt >> 2) + row) + (mt & 3) if quad else 16 * mt + row
                            block[kidx * NC + nt * 16 + I[l]] += acc[mt, nt, l, r]
            for nt in range(NT):
                for i in range(16):
                    block[KN + nt * 16 + i] += bsum[nt, i] + bsum[nt, i + 16] + bsum[nt, i + 32] + bsum[nt, i + 48]
        ws[ms * PER:(ms + 1) * PER] = block
    # pass 2
    tot = ws[:msplits * PER].reshape(msplits, PER).astype(np.float64).sum(0)
    dw = np.zeros(taps * cin * N, np.float64)
    db = np.zeros(cout, np.float64)
    for idx in range(KN):
        kidx, ncol = idx // NC, idx % NC
        if kidx < K and ncol < N:
            dw[keras_widx(mode, kidx // cin, kidx % cin, ncol, cin, cout)] += tot[idx]
    for ncol in range(N):
        db[ncol % cout if mode == DECONV_K2S2 else ncol] += tot[KN + ncol]
    return dw, db


@pytest.mark.parametrize('mode,n,h,w,c0,c1,cout,rows,quad', [
    (CONV_K2S1, 1, 5, 6, 16, 0, 16, 16, False), (CONV_K2S1, 1, 5, 6, 16, 0, 16, 16, True),
    (CONV_K2S2, 1, 4, 8, 32, 0, 32, 32, True), (CONV_K2S2, 1, 4, 8, 16, 0, 32, 16, False),
    (DECONV_K2S2, 1, 3, 4, 16, 64, 8, 16, True), (DECONV_K2S2, 1, 3, 4, 8, 32, 4, 32, False),
    (DECONV_K2S1, 2, 3, 4, 8, 0, 8, 16, False), (DECONV_K2S1, 1, 4, 4, 16, 0, 16, 16, True),
    (CONV_K2S1, 1, 3, 5, 5, 0, 7, 16, False), (CONV_K2S2, 1, 4, 4, 3, 6, 12, 16, False),
])
def test_wgrad_narrow_kernel_index_math(mode, n, h, w, c0, c1, cout, rows, quad):
    rng = np.random.default_rng(mode * 7 + cout + int(quad))
    tr = mode in (DECONV_K2S2, DECONV_K2S1)
    s = 2 if mode in (CONV_K2S2, DECONV_K2S2) else 1
    cin = c0 + c1
    pad0, pad1, padp = 4, 8, 4
    x0 = rng.standard_normal((n, h, w, c0 + pad0)).astype(np.float32)
    x1 = rng.standard_normal((n, h, w, c1 + pad1)).astype(np.float32)
    x = np.concatenate((x0[..., :c0], x1[..., :c1]), -1) if c1 else x0[..., :c0]
    wshape = (2, 2, cout, cin) if tr else (2, 2, cin, cout)
    wz = torch.zeros(wshape, requires_grad=True); bz = torch.zeros(cout, requires_grad=True)
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    y = f(torch.tensor(x), wz, bz, s)
    dp = rng.standard_normal(tuple(y.shape[:3]) + (cout + padp,)).astype(np.float32)
    gw, gb = torch.autograd.grad(y, (wz, bz), torch.tensor(dp[..., :cout]))
    dw, db = emulate_wgrad_narrow(mode, x0.reshape(-1), c0 + pad0, c0, x1.reshape(-1), c1 + pad1, c1, n, h, w,
                                  dp.reshape(-1), cout + padp, cout, rows, quad)
    np.testing.assert_allclose(dw.reshape(wshape), gw.numpy(), atol=1e-4)
    np.testing.assert_allclose(db, gb.numpy(), atol=1e-4)
