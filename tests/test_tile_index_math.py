"""CPU: lane-level NumPy emulation of csrc/conv_tile.hip (weight packing order, per-thread copy units of the
A / B slabs, planar LDS layouts incl. the x-parity split of the stride-2 conv, fragment read offsets, stage /
observation sequencing, epilogue addressing and the in-register observation mean), compared with the oracle's
Conv2D 'same' + LeakyReLU.  Mirrors the kernel statement by statement (v_mfma_f32_16x16x4_f32 layout as in
tests/test_mfma_index_math.py)."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T

K2S2, K2S1 = 1, 2
TH, TW, PL = 8, 16, 160
LANE = np.arange(64); KK, J = LANE >> 4, LANE & 15


def mfma(a, b, acc):
    A = np.zeros((16, 4), np.float32); B = np.zeros((4, 16), np.float32)
    A[J, KK] = a; B[KK, J] = b
    D = A @ B
    for r in range(4):
        acc[:, r] += D[KK * 4 + r, J]
    return acc


def pack(wk, cin, cout, tnt):
    total = 4 * cin * cout
    wp = np.zeros(total, np.float32)
    flat = wk.reshape(-1)
    ncc = cin >> 4
    for idx in range(total):
        s4, lane = idx & 3, (idx >> 2) & 63
        r = idx >> 8
        ct = r % tnt; r //= tnt
        t = r & 3; r >>= 2
        cc = r % ncc; g = r // ncc
        c = cc * 16 + 4 * (lane >> 4) + s4
        o = (g * tnt + ct) * 16 + (lane & 15)
        wp[idx] = flat[(t * cin + c) * cout + o]
    return wp


def xcd_tile(b, nblocks):
    return b if nblocks & 7 else (b & 7) * (nblocks >> 3) + (b >> 3)


def conv_tile(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, tn, ldo, ldm, act, alpha, want_mean):
    tnt = tn // 16
    WN = 2 if tnt >= 4 else 1
    WM = 4 // WN; RT = TH // WM; CT = tnt // WN
    stage_taps = 4 if mode == K2S1 else 2
    b_units = 153 * 4 if mode == K2S1 else 1024
    b_floats = 4 * PL * 4 if mode == K2S1 else 4 * 2 * 128 * 4
    a_floats = stage_taps * tnt * 256
    NA, NB = a_floats // 4 // 256, (b_units + 255) // 256
    stage = a_floats + b_floats
    oh, ow = (h // 2, w // 2) if mode == K2S2 else (h, w)
    tiles_y, tiles_x = (oh + TH - 1) // TH, (ow + TW - 1) // TW
    ncc = cin // 16
    S = src.reshape(-1)
    out = np.full(frames * kobs * oh * ow * ldo, np.nan, np.float32)
    mean_out = np.full(frames * oh * ow * ldm, np.nan, np.float32) if want_mean else None
    ntiles = frames * tiles_y * tiles_x
    spf = (1 if mode == K2S1 else 2) * ncc
    total_stages = spf * kobs
    in_frame = h * w
    for bx in range(ntiles):
        for g in range(cout // tn):
            tile = xcd_tile(bx, ntiles)
            tx0 = (tile % tiles_x) * TW; tile //= tiles_x
            ty0 = (tile % tiles_y) * TH
            f = tile // tiles_y
            lds = np.full(2 * stage, np.nan, np.float32)
            tid = np.arange(256)
            b_lds, b_tex, b_ok = [], [], []
            for i in range(NB):
                u = tid + 256 * i
                q, tx = u & 3, u >> 2
                if mode == K2S1:
                    hy, hx = tx // 17, tx % 17
                    gy, gx = ty0 + hy, tx0 + hx
                    b_ok.append((u < b_units) & (gy < h) & (gx < w)); b_tex.append(gy * w + gx); b_lds.append((q * PL + tx) * 4)
                else:
                    y, xx = tx >> 5, tx & 31
                    gy, gx = 2 * (ty0 + y), 2 * tx0 + xx
                    b_ok.append(((ty0 + y) < oh) & (gx < w)); b_tex.append(gy * w + gx)
                    b_lds.append(((q * 2 + (xx & 1)) * 128 + y * 16 + (xx >> 1)) * 4)

            def load_stage(q):
                i = q // spf; s = q - i * spf
                cc = s if mode == K2S1 else s >> 1
                a = 0 if mode == K2S1 else s & 1
                ap = (((g * ncc + cc) * 4 + 2 * a) * tnt) * 256
                ra = [np.stack([packed[ap + (tid + 256 * n) * 4 + e] for e in range(4)], 1) for n in range(NA)]
                sp = ((f * kobs + i) * in_frame + a * w) * ld + cc * 16
                rb = []
                for n in range(NB):
                    u = tid + 256 * n
                    addr = sp + np.where(b_ok[n], b_tex[n], 0) * ld + 4 * (u & 3)
                    v = np.stack([S[np.minimum(addr + e, len(S) - 1)] for e in range(4)], 1)
                    v[~b_ok[n]] = 0
                    rb.append(v)
                return ra, rb

            def store_stage(buf, ra, rb):
                base = buf * stage
                for n in range(NA):
                    for t_ in range(256):
                        a0 = base + (t_ + 256 * n) * 4
                        lds[a0:a0 + 4] = ra[n][t_]
                for n in range(NB):
                    for t_ in range(256):
                        if t_ + 256 * n < b_units:
                            a0 = base + a_floats + b_lds[n][t_]
                            lds[a0:a0 + 4] = rb[n][t_]

            acc = [[[np.zeros((64, 4), np.float32) for _ in range(CT)] for _ in range(RT)] for _ in range(4)]
            mean = [[[np.zeros((64, 4), np.float32) for _ in range(CT)] for _ in range(RT)] for _ in range(4)]
            ra, rb = load_stage(0); store_stage(0, ra, rb)
            for q in range(total_stages):
                if q + 1 < total_stages:
                    ra, rb = load_stage(q + 1)
                A0 = (q & 1) * stage; B0 = A0 + a_floats
                s = q % spf
                for wave in range(4):
                    wn, wm = wave % WN, wave // WN
                    for tl in range(stage_taps):
                        bf, af = [], []
                        for rt in range(RT):
                            y = wm * RT + rt
                            off = (KK * PL + (y + (tl >> 1)) * 17 + J + (tl & 1)) * 4 if mode == K2S1 else \
                                ((KK * 2 + tl) * 128 + y * 16 + J) * 4
                            v = np.stack([lds[B0 + off + e] for e in range(4)], 1)
                            assert not np.isnan(v).any()
                            bf.append(v)
                        for ct in range(CT):
                            off = ((tl * tnt + wn * CT + ct) * 64 + LANE) * 4
                            v = np.stack([lds[A0 + off + e] for e in range(4)], 1)
                            assert not np.isnan(v).any()
                            af.append(v)
                        for s4 in range(4):
                            for rt in range(RT):
                                for ct in range(CT):
                                    acc[wave][rt][ct] = mfma(af[ct][:, s4], bf[rt][:, s4], acc[wave][rt][ct])
                    if s == spf - 1:
                        i = q // spf
                        for ct in range(CT):
                            oc = (g * tnt + wn * CT + ct) * 16 + 4 * KK
                            bv = np.stack([bias[oc + e] for e in range(4)], 1)
                            for rt in range(RT):
                                gy = ty0 + wm * RT + rt; gx = tx0 + J
                                v = acc[wave][rt][ct] + bv
                                if act:
                                    v = np.where(v > 0, v, np.float32(alpha) * v).astype(np.float32)
                                acc[wave][rt][ct] = np.zeros((64, 4), np.float32)
                                mean[wave][rt][ct] = mean[wave][rt][ct] + v
                                for l in np.nonzero((gy < oh) & (gx < ow))[0]:
                                    ot = ((f * kobs + i) * oh + gy) * ow + gx[l]
                                    a0 = ot * ldo + oc[l]
                                    assert np.isnan(out[a0])
                                    out[a0:a0 + 4] = v[l]
                                    if want_mean and i == kobs - 1:
                                        mt = (f * oh + gy) * ow + gx[l]
                                        m0 = mt * ldm + oc[l]
                                        mean_out[m0:m0 + 4] = mean[wave][rt][ct][l] * np.float32(1.0 / kobs)
                if q + 1 < total_stages:
                    store_stage((q + 1) & 1, ra, rb)
    return out, mean_out


@pytest.mark.parametrize('mode,cin,cout,tn,h,w,kobs', [(K2S1, 16, 32, 32, 10, 20, 2), (K2S1, 32, 64, 64, 8, 16, 1),
                                                        (K2S2, 16, 32, 32, 20, 36, 2), (K2S2, 32, 128, 64, 16, 32, 1),
                                                        ])
def test_conv_tile_emulation_matches_oracle(mode, cin, cout, tn, h, w, kobs):
    rng = np.random.default_rng(cin + cout + h)
    frames = 2
    ld = cin + 4                                  # a channel slice of a wider tensor
    src = rng.standard_normal((frames * kobs, h, w, ld)).astype(np.float32)
    wk = (rng.standard_normal((2, 2, cin, cout)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32) * 0.1
    oh, ow = (h // 2, w // 2) if mode == K2S2 else (h, w)
    ldo, ldm = cout, 2 * cout
    out, mean = conv_tile(mode, src, ld, cin, frames, kobs, h, w, pack(wk, cin, cout, tn // 16), bias, cout, tn, ldo, ldm,
                          True, 0.3, True)
    with torch.no_grad():
        ref = T.leaky_relu(T.conv2d_same(torch.from_numpy(src[..., :cin].copy()), torch.from_numpy(wk), torch.from_numpy(bias),
                                         2 if mode == K2S2 else 1), 0.3).numpy()
    out = out.reshape(frames * kobs, oh, ow, ldo)
    assert not np.isnan(out).any()
    assert np.abs(out - ref).max() < 1e-4
    m = mean.reshape(frames, oh, ow, ldm)[..., :cout]
    assert np.abs(m - ref.reshape(frames, kobs, oh, ow, cout).mean(1)).max() < 1e-4


@pytest.mark.parametrize('h,w,cout,tn', [(8, 16, 16, 64), (9, 17, 16, 32), (20, 12, 32, 64), (5, 40, 8, 32), (16, 32, 16, 64)])
def test_transposed_k2s2_mode_tiles_the_input_grid_and_writes_every_output_texel_once(h, w, cout, tn):
    """NLT_DECONV_K2S2 (backward-data of a stride-2 conv = Conv2DTranspose k2s2): the workgroup grid covers the INPUT grid (h x w) in
    8 x 16 tiles -- nlt_conv_tile_backward_data launched 4x that before r04 (tiles from the 2h x 2w output) -- and the GEMM's 4 * cout
    columns are (ab, o): row tile (g, wn, ct), lane group kk -> column 16 (g TNT + wn CT + ct) + 4 kk -> (ab, oc), output texel
    (2 y + ab / 2, 2 x + ab % 2).  Every (output texel, channel quad) must be written by exactly one (workgroup, wave, lane, rt, ct)."""
    tnt = tn // 16
    WN = 2 if tnt >= 4 else 1
    WM = 4 // WN; RT = TH // WM; CT = tnt // WN
    ncols = 4 * cout
    assert ncols % tn == 0 or ncols < tn
    groups = (ncols + tn - 1) // tn
    tiles_y, tiles_x = (h + TH - 1) // TH, (w + TW - 1) // TW            # input grid (the fixed rule)
    oh, ow = 2 * h, 2 * w
    seen = np.zeros((oh, ow, cout // 4), np.int64)
    for g in range(groups):
        for ty in range(tiles_y):
            for tx in range(tiles_x):
                ty0, tx0 = ty * TH, tx * TW
                for wave in range(4):
                    wn, wm = wave % WN, wave // WN
                    for ct in range(CT):
                        for kk in range(4):
                            col = (g * tnt + wn * CT + ct) * 16 + 4 * kk
                            if col >= ncols:
                                continue
                            ab, oc = divmod(col, cout)
                            assert oc % 4 == 0 and ab < 4
                            for rt in range(RT):
                                for j in range(16):
                                    gy, gx = 2 * (ty0 + wm * RT + rt) + (ab >> 1), 2 * (tx0 + j) + (ab & 1)
                                    if gy < oh and gx < ow:
                                        seen[gy, gx, oc // 4] += 1
    assert (seen == 1).all()
    # the pre-r04 grid (tiles of the OUTPUT dims) would have launched (about) four times the workgroups for the same writes
    assert ((oh + TH - 1) // TH) * ((ow + TW - 1) // TW) >= 2 * tiles_y * tiles_x
