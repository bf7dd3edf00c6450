"""CPU: lane-level emulation of csrc/conv_mfma.hip's index arithmetic (weight packing, permuted-k
fragment loads, segment/tap order, DECONV column mapping, epilogue addressing) under the
documented v_mfma_f32_16x16x4_f32 fragment layout (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
D[row=(l>>4)*4+r][col=l&15]; cdna_hip_programming.md 3).  Mirrors the kernel line by line in
NumPy so a logic slip is caught before spending GPU time; the hardware layout itself is what
the -m gpu tests confirm."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T

CONV1X1, CONV_K2S2, CONV_K2S1, DECONV_K2S2, DECONV_K2S1 = range(5)
TAPS = {0: 1, 1: 4, 2: 4, 3: 1, 4: 4}


def keras_widx(mode, t, c, ncol, cin, cout):
    if mode in (CONV1X1, CONV_K2S2, CONV_K2S1):
        return (t * cin + c) * cout + ncol
    if mode == DECONV_K2S1:
        return (t * cout + ncol) * cin + c
    return ncol * cin + c


def chunks16(c):
    return (c + 15) >> 4


def pack(mode, wk, c0, c1, cout):
    N = 4 * cout if mode == DECONV_K2S2 else cout
    ntiles = (N + 15) >> 4
    ch0, ch1 = chunks16(c0), chunks16(c1)
    total = TAPS[mode] * (ch0 + ch1) * ntiles * 256
    wp = np.zeros(total, np.float32)
    flat = wk.reshape(-1)
    for idx in range(total):
        s4 = idx & 3; lane = (idx >> 2) & 63; tile = idx >> 8
        nt = tile % ntiles; kc = tile // ntiles
        t = kc // (ch0 + ch1); r = kc % (ch0 + ch1)
        s = r >= ch0
        cl = ((r - ch0) if s else r) * 16 + 4 * (lane >> 4) + s4
        cs = c1 if s else c0
        ncol = nt * 16 + (lane & 15)
        if cl < cs and ncol < N:
            wp[idx] = flat[keras_widx(mode, t, (c0 if s else 0) + cl, ncol, c0 + c1, cout)]
    return wp, ntiles


def tap_texel(mode, h, w, f, y, x, t):
    a, b = t >> 1, t & 1
    if mode in (CONV1X1, DECONV_K2S2):
        iy, ix = y, x
    elif mode == CONV_K2S2:
        iy, ix = 2 * y + a, 2 * x + b
    elif mode == CONV_K2S1:
        iy, ix = y + a, x + b
        if iy >= h or ix >= w:
            return -1
    else:
        iy, ix = y - a, x - b
        if iy < 0 or ix < 0:
            return -1
    return (f * h + iy) * w + ix


def mfma(a, b, acc):
    A = np.zeros((16, 4), np.float32); B = np.zeros((4, 16), np.float32)
    for l in range(64):
        A[l & 15, l >> 4] = a[l]; B[l >> 4, l & 15] = b[l]
    Dm = A @ B
    for l in range(64):
        for r in range(4):
            acc[l, r] += Dm[(l >> 4) * 4 + r, l & 15]


def emulate(mode, src0, ld0, c0, src1, ld1, c1, n, h, w, wp, ntiles, bias, cout, ldo, RT, CT, act, alpha):
    gh, gw, oh, ow, N = h, w, h, w, cout
    if mode == CONV_K2S2:
        gh = oh = h // 2; gw = ow = w // 2
    if mode == DECONV_K2S2:
        oh, ow, N = 2 * h, 2 * w, 4 * cout
    M = n * gh * gw
    out = np.full(n * oh * ow * ldo, np.nan, np.float32)
    ngroups = ntiles // CT
    mtiles = (M + 16 * RT - 1) // (16 * RT)
    for wave in range(mtiles * ngroups):
        ng, mt = wave % ngroups, wave // ngroups
        acc = np.zeros((RT, CT, 64, 4), np.float32)
        rows = {}
        for rt in range(RT):
            for lane in range(64):
                m = (mt * RT + rt) * 16 + (lane & 15)
                rv = m < M
                mc = m if rv else M - 1
                rows[rt, lane] = (rv, mc, mc % gw, (mc // gw) % gh, mc // (gw * gh))
        kc = 0
        for t in range(TAPS[mode]):
            for s in range(2):
                cs = c1 if s else c0
                if cs == 0:
                    continue
                src, ld = (src1, ld1) if s else (src0, ld0)
                for k0 in range(0, cs, 16):
                    bfr = np.zeros((RT, 64, 4), np.float32)
                    afr = np.zeros((CT, 64, 4), np.float32)
                    for lane in range(64):
                        kk = lane >> 4
                        kin = (k0 + 4 * kk) < cs
                        for rt in range(RT):
                            rv, mc, x, y, f = rows[rt, lane]
                            tex = tap_texel(mode, h, w, f, y, x, t) if rv else -1
                            if kin and tex >= 0:
                                o = tex * ld + 4 * kk + k0
                                bfr[rt, lane] = src[o:o + 4]
                        for ct in range(CT):
                            o = ((kc * ntiles + ng * CT + ct) * 64 + lane) * 4
                            afr[ct, lane] = wp[o:o + 4]
                    for s4 in range(4):
                        for rt in range(RT):
                            for ct in range(CT):
                                mfma(afr[ct, :, s4], bfr[rt, :, s4], acc[rt, ct])
                    kc += 1
        for ct in range(CT):
            for lane in range(64):
                kk = lane >> 4
                ncol = (ng * CT + ct) * 16 + kk * 4
                if ncol >= N:
                    continue
                oc, ab = ncol, 0
                if mode == DECONV_K2S2:
                    ab = ncol // cout; oc = ncol - ab * cout
                for rt in range(RT):
                    rv, mc, x, y, f = rows[rt, lane]
                    if not rv:
                        continue
                    otex = mc
                    if mode == DECONV_K2S2:
                        otex = (f * oh + 2 * y + (ab >> 1)) * ow + 2 * x + (ab & 1)
                    v = acc[rt, ct, lane] + bias[oc:oc + 4]
                    if act:
                        v = np.where(v > 0, v, alpha * v)
                    o = otex * ldo + oc
                    assert np.all(np.isnan(out[o:o + 4])), "two lanes wrote the same output"
                    out[o:o + 4] = v
    return out.reshape(n, oh, ow, ldo)


@pytest.mark.parametrize('mode,n,h,w,c0,c1,cout,RT,CT', [
    (CONV1X1, 1, 3, 5, 16, 0, 16, 1, 1),
    (CONV_K2S2, 1, 4, 6, 16, 0, 32, 2, 2),
    (CONV_K2S1, 2, 3, 3, 8, 4, 16, 1, 1),
    (DECONV_K2S2, 1, 2, 3, 8, 32, 4, 1, 1),
    (DECONV_K2S2, 1, 3, 2, 16, 0, 16, 2, 4),
    (DECONV_K2S1, 1, 3, 4, 4, 0, 4, 2, 1),
    (CONV1X1, 1, 2, 2, 4, 32, 12, 1, 1),
])
def test_mfma_kernel_index_math(mode, n, h, w, c0, c1, cout, RT, CT):
    rng = np.random.default_rng(mode * 7 + cout)
    tr = mode in (DECONV_K2S2, DECONV_K2S1)
    k = 1 if mode == CONV1X1 else 2
    s = 2 if mode in (CONV_K2S2, DECONV_K2S2) else 1
    cin = c0 + c1
    pad0, pad1, pado = 4, 8, 4
    x0 = rng.standard_normal((n, h, w, c0 + pad0)).astype(np.float32)
    x1 = rng.standard_normal((n, h, w, c1 + pad1)).astype(np.float32)
    wk = rng.standard_normal((k, k, cout, cin) if tr else (k, k, cin, cout)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    wp, ntiles = pack(mode, wk, c0, c1, cout)
    got = emulate(mode, x0.reshape(-1), c0 + pad0, c0, x1.reshape(-1), c1 + pad1, c1, n, h, w, wp, ntiles, b,
                  cout, cout + pado, RT, CT, True, 0.3)
    x = np.concatenate((x0[..., :c0], x1[..., :c1]), -1) if c1 else x0[..., :c0]
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    ref = T.leaky_relu(f(torch.tensor(x), torch.tensor(wk), torch.tensor(b), s), 0.3).numpy()
    np.testing.assert_allclose(got[..., :cout], ref, atol=1e-4)
    assert np.all(np.isnan(got[..., cout:]))
