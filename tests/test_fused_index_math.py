"""CPU: lane-level NumPy emulation of csrc/fused.hip (weight folding/packing, haloed-tile flattening,
tap <-> lane-group mapping, LDS hand-off, epilogue addressing of the front and back kernels) under the
v_mfma_f32_16x16x4_f32 fragment layout (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)*4+r][col=l&15])
that the -m gpu conv tests already confirmed on hardware.  Mirrors the kernels statement by statement so
an indexing slip is caught before GPU time is spent; compared with the oracle's layer-by-layer result."""
import numpy as np
import pytest
import torch

from oracle import nlt_oracle as O
from oracle import tf_ops as T

TH, TW = 8, 16
HH, HW = TH + 1, TW + 1
HT = HH * HW
NT = (HT + 15) // 16
PLT = 160
PATH = 4 * PLT * 4
OFF_AQ2, OFF_AO2, OFF_AQ1, OFF_AO1 = 0, 512, 704, 1728
OFF_BQ2, OFF_BO2, OFF_BQ1, OFF_BO1, OFF_WSK, OFF_BSK, BLOB = 2752, 2768, 2784, 2800, 2816, 2840, 2848
LANE = np.arange(64)
KK, J = LANE >> 4, LANE & 15


def mfma(a, b, acc):
    """acc[lane, r] += D[(lane>>4)*4 + r][lane&15] with D = A @ B."""
    A = np.zeros((16, 4), np.float32); B = np.zeros((4, 16), np.float32)
    A[J, KK] = a; B[KK, J] = b
    D = A @ B
    for r in range(4):
        acc[:, r] += D[KK * 4 + r, J]
    return acc


def lrelu(v, alpha):
    return np.where(v > 0, v, alpha * v).astype(np.float32)


def pack(w):
    """front_pack_kernel, one blob float at a time."""
    f = {k: np.asarray(v, np.float32).reshape(-1) for k, v in w.items()}
    blob = np.zeros(BLOB, np.float32)
    for idx in range(BLOB):
        v = 0.0
        if idx < OFF_AO2:
            m, lane = idx >> 6, idx & 63; tap, o = lane >> 4, lane & 15
            for c in range(16):
                v += f['wq0'][m * 16 + c] * f['wqa'][((tap * 32) + c) * 16 + o] if m < 5 else \
                    f['wo0'][(m - 5) * 16 + c] * f['wqa'][((tap * 32) + 16 + c) * 16 + o]
        elif idx < OFF_AQ1:
            r = idx - OFF_AO2; m, lane = r >> 6, r & 63; tap, o = lane >> 4, lane & 15
            for c in range(16):
                v += f['wo0'][m * 16 + c] * f['woa'][((tap * 16) + c) * 16 + o]
        elif idx < OFF_BQ2:
            obs = idx >= OFF_AO1
            r = idx - (OFF_AO1 if obs else OFF_AQ1)
            s4, lane, tap = r & 3, (r >> 2) & 63, r >> 8
            v = (f['wob'] if obs else f['wqb'])[((tap * 16) + 4 * (lane >> 4) + s4) * 16 + (lane & 15)]
        elif idx < OFF_BO2:
            o = idx - OFF_BQ2; v = f['bqa'][o]
            for t in range(4):
                for c in range(16):
                    v += f['bq0'][c] * f['wqa'][((t * 32) + c) * 16 + o] + f['bo0'][c] * f['wqa'][((t * 32) + 16 + c) * 16 + o]
        elif idx < OFF_BQ1:
            o = idx - OFF_BO2; v = f['boa'][o]
            for t in range(4):
                for c in range(16):
                    v += f['bo0'][c] * f['woa'][((t * 16) + c) * 16 + o]
        elif idx < OFF_BO1:
            v = f['bqb'][idx - OFF_BQ1]
        elif idx < OFF_WSK:
            v = f['bob'][idx - OFF_BO1]
        elif idx < OFF_BSK:
            r, o = divmod(idx - OFF_WSK, 3)
            for c in range(16):
                v += f['wq0'][r * 16 + c] * f['wh'][(4 + c) * 3 + o] if r < 5 else f['wo0'][(r - 5) * 16 + c] * f['wh'][(20 + c) * 3 + o]
        elif idx < OFF_BSK + 3:
            o = idx - OFF_BSK; v = f['bh'][o]
            for c in range(16):
                v += f['bq0'][c] * f['wh'][(4 + c) * 3 + o] + f['bo0'][c] * f['wh'][(20 + c) * 3 + o]
        blob[idx] = v
    return blob


def xcd_tile(b, nblocks):
    return b if nblocks & 7 else (b & 7) * (nblocks >> 3) + (b >> 3)


def front(base, cvis, lvis, nn_rgb, nn_base, blob, add_base, alpha, blob3=None):
    n, h, w, _ = base.shape
    k = nn_rgb.shape[1]
    h2, w2 = h // 2, w // 2
    ty, tx = (h2 + TH - 1) // TH, (w2 + TW - 1) // TW
    nblocks = n * ty * tx
    fm1 = np.full((n * h2 * w2 * 32,), np.nan, np.float32)
    obs1 = np.full((n * k * h2 * w2 * 16,), np.nan, np.float32)
    skip3 = np.full((n * h * w * 3,), np.nan, np.float32)
    h4, w4 = h2 // 2, w2 // 2
    qtmp2 = np.full((n * h4 * w4 * 32,), np.nan, np.float32)
    otmp2 = np.full((n * k * h4 * w4 * 32,), np.nan, np.float32)
    B_, C_, L_, R_, NB_ = (a.reshape(-1) for a in (base, cvis, lvis, nn_rgb, nn_base))
    hw, hw2 = h * w, h2 * w2
    inv_k = np.float32(1.0 / k)
    seen = set()
    for blk in range(nblocks):
        tile = xcd_tile(blk, nblocks); seen.add(tile)
        tx0 = (tile % tx) * TW; tile //= tx
        ty0 = (tile % ty) * TH
        f = tile // ty
        lds = np.full((max((1 + k) * PATH, (2 + k) * 2048),), np.nan, np.float32)
        keep = {}                                              # L2S2: (wave, e) -> (qv, mean, [o1_i]) kept in registers
        for wave in range(4):
            aq2 = [blob[OFF_AQ2 + m * 64 + LANE] for m in range(8)]
            ao2 = [blob[OFF_AO2 + m * 64 + LANE] for m in range(3)]
            bq2 = np.stack([blob[OFF_BQ2 + 4 * KK + r] for r in range(4)], 1)
            bo2 = np.stack([blob[OFF_BO2 + 4 * KK + r] for r in range(4)], 1)
            for mt in range(wave, NT, 4):
                t = mt * 16 + J
                live = t < HT
                hy = np.where(live, t // HW, 0); hx = np.where(live, t % HW, 0)
                gy, gx = ty0 + hy, tx0 + hx
                inside = live & (gy < h2) & (gx < w2)
                owned = inside & (hy < TH) & (hx < TW)
                fy = np.where(inside, 2 * gy + (KK >> 1), 0); fx = np.where(inside, 2 * gx + (KK & 1), 0)
                tex = f * hw + fy * w + fx
                raw = [B_[tex * 3], B_[tex * 3 + 1], B_[tex * 3 + 2], C_[tex], L_[tex]]
                xs = [np.zeros(64, np.float32) for _ in range(3)]
                for i in range(k):
                    ot = (f * k + i) * hw + fy * w + fx
                    d = [R_[ot * 3 + c] - NB_[ot * 3 + c] for c in range(3)]
                    for c in range(3):
                        xs[c] = xs[c] + d[c]
                    acc = np.zeros((64, 4), np.float32)
                    for m in range(3):
                        acc = mfma(ao2[m], d[m], acc)
                    acc = lrelu(acc + bo2, alpha)
                    acc[~inside] = 0
                    for l in np.nonzero(live)[0]:
                        a0 = (1 + i) * PATH + (KK[l] * PLT + t[l]) * 4
                        lds[a0:a0 + 4] = acc[l]
                raw += [xs[c] * inv_k for c in range(3)]
                acc = np.zeros((64, 4), np.float32)
                for m in range(8):
                    acc = mfma(aq2[m], raw[m], acc)
                acc = lrelu(acc + bq2, alpha)
                acc[~inside] = 0
                for l in np.nonzero(live)[0]:
                    a0 = (KK[l] * PLT + t[l]) * 4
                    lds[a0:a0 + 4] = acc[l]
                for l in np.nonzero(owned)[0]:
                    s = [blob[OFF_BSK + o] for o in range(3)]
                    for r in range(8):
                        for o in range(3):
                            s[o] = s[o] + raw[r][l] * blob[OFF_WSK + r * 3 + o]
                    if add_base:
                        s = [s[o] + raw[o][l] for o in range(3)]
                    assert np.isnan(skip3[tex[l] * 3])                 # every texel written exactly once
                    skip3[tex[l] * 3: tex[l] * 3 + 3] = s
        # __syncthreads()
        for wave in range(4):
            aq1 = [np.stack([blob[OFF_AQ1 + (t_ * 64 + LANE) * 4 + s4] for s4 in range(4)], 1) for t_ in range(4)]
            ao1 = [np.stack([blob[OFF_AO1 + (t_ * 64 + LANE) * 4 + s4] for s4 in range(4)], 1) for t_ in range(4)]
            bq1 = np.stack([blob[OFF_BQ1 + 4 * KK + r] for r in range(4)], 1)
            bo1 = np.stack([blob[OFF_BO1 + 4 * KK + r] for r in range(4)], 1)
            for r in range(wave, TH, 4):
                gy, gx = ty0 + r, tx0 + J
                inside = (gy < h2) & (gx < w2)
                otex = gy * w2 + gx
                mean = np.zeros((64, 4), np.float32); qv = None
                o1s = []
                for p in range(k + 1):
                    acc = np.zeros((64, 4), np.float32)
                    for t_ in range(4):
                        a0 = p * PATH + (KK * PLT + (r + (t_ >> 1)) * HW + J + (t_ & 1)) * 4
                        b = np.stack([lds[a0 + s4] for s4 in range(4)], 1)
                        assert not np.isnan(b).any()
                        a = ao1[t_] if p else aq1[t_]
                        for s4 in range(4):
                            acc = mfma(a[:, s4], b[:, s4], acc)
                    acc = lrelu(acc + (bo1 if p else bq1), alpha)
                    if p == 0:
                        qv = acc
                    else:
                        mean = mean + acc
                        o1s.append(acc)
                        for l in np.nonzero(inside)[0]:
                            a0 = ((f * k + (p - 1)) * hw2 + otex[l]) * 16 + 4 * KK[l]
                            obs1[a0:a0 + 4] = acc[l]
                mean = mean * inv_k
                keep[(wave, r)] = (qv, mean, o1s)
                for l in np.nonzero(inside)[0]:
                    a0 = (f * hw2 + otex[l]) * 32 + 4 * KK[l]
                    fm1[a0:a0 + 4] = qv[l]
                    fm1[a0 + 16:a0 + 20] = mean[l]
        if blob3 is not None:
            # __syncthreads(); level-1 tile -> LDS [slab][quad][x parity][8][8][4], swizzled; __syncthreads(); level 2's stride-2 convs
            lds[:] = np.nan
            for (wave, r), (qv, mean, o1s) in keep.items():
                for l in range(64):
                    par = J[l] & 1
                    d0 = (KK[l] * 128 + par * 64 + ((r * 8 + (J[l] >> 1)) ^ ((((r >> 1) & 1) ^ par) * 8))) * 4
                    lds[d0:d0 + 4] = qv[l]; lds[d0 + 2048:d0 + 2052] = mean[l]
                    for i, o in enumerate(o1s):
                        lds[d0 + (2 + i) * 2048: d0 + (2 + i) * 2048 + 4] = o[l]
            for wave in range(4):
                ct, rt = wave & 1, wave >> 1
                t2 = ct * 16 + J
                Y, X = t2 >> 3, t2 & 7
                src = ((KK & 1) * 64 + (((2 * Y + (KK >> 1)) * 8 + X) ^ (((Y & 1) ^ (KK & 1)) * 8))) * 4
                gy2, gx2 = (ty0 >> 1) + Y, (tx0 >> 1) + X
                in2 = (gy2 < h4) & (gx2 < w4)
                tex2 = gy2 * w4 + gx2
                oc = rt * 16 + 4 * KK
                acc = np.zeros((64, 4), np.float32)
                for c8 in range(8):
                    v = np.stack([lds[src + (c8 >> 2) * 2048 + 512 * (c8 & 3) + e] for e in range(4)], 1)
                    assert not np.isnan(v).any()
                    a = np.stack([blob3[OFF3_AQ + ((rt * 8 + c8) * 64 + LANE) * 4 + e] for e in range(4)], 1)
                    for e in range(4):
                        acc = mfma(a[:, e], v[:, e], acc)
                acc = lrelu(acc + np.stack([blob3[OFF3_BQ + oc + e] for e in range(4)], 1), alpha)
                for l in np.nonzero(in2)[0]:
                    a0 = (f * h4 * w4 + tex2[l]) * 32 + oc[l]
                    assert np.isnan(qtmp2[a0]); qtmp2[a0:a0 + 4] = acc[l]
                for i in range(k):
                    acc = np.zeros((64, 4), np.float32)
                    for c4 in range(4):
                        v = np.stack([lds[src + (2 + i) * 2048 + 512 * c4 + e] for e in range(4)], 1)
                        assert not np.isnan(v).any()
                        a = np.stack([blob3[OFF3_AO + ((rt * 4 + c4) * 64 + LANE) * 4 + e] for e in range(4)], 1)
                        for e in range(4):
                            acc = mfma(a[:, e], v[:, e], acc)
                    acc = lrelu(acc + np.stack([blob3[OFF3_BO + oc + e] for e in range(4)], 1), alpha)
                    for l in np.nonzero(in2)[0]:
                        a0 = ((f * k + i) * h4 * w4 + tex2[l]) * 32 + oc[l]
                        assert np.isnan(otmp2[a0]); otmp2[a0:a0 + 4] = acc[l]
    assert seen == set(range(nblocks))
    out = (fm1.reshape(n, h2, w2, 32), obs1.reshape(n, k, h2, w2, 16), skip3.reshape(n, h, w, 3))
    if blob3 is not None:
        out += (qtmp2.reshape(n, h4, w4, 32), otmp2.reshape(n, k, h4, w4, 32))
    return out


OFF3_AQ, OFF3_AO, OFF3_BQ, OFF3_BO, BLOB3 = 0, 4096, 6144, 6176, 6208


def pack_l2(wq, bq, wo, bo):
    """front_pack_l2_kernel."""
    wq, bq, wo, bo = (np.asarray(a, np.float32).reshape(-1) for a in (wq, bq, wo, bo))
    blob = np.zeros(BLOB3, np.float32)
    for idx in range(BLOB3):
        if idx < OFF3_AO:
            e, lane, c8, rt = idx & 3, (idx >> 2) & 63, (idx >> 8) & 7, idx >> 11
            tap, o, c = lane >> 4, rt * 16 + (lane & 15), 16 * (c8 >> 2) + 4 * (c8 & 3) + e
            blob[idx] = wq[((tap * 32) + c) * 32 + o]
        elif idx < OFF3_BQ:
            r = idx - OFF3_AO
            e, lane, c4, rt = r & 3, (r >> 2) & 63, (r >> 8) & 3, r >> 10
            blob[idx] = wo[(((lane >> 4) * 16) + 4 * c4 + e) * 32 + rt * 16 + (lane & 15)]
        elif idx < OFF3_BO:
            blob[idx] = bq[idx - OFF3_BQ]
        else:
            blob[idx] = bo[idx - OFF3_BO]
    return blob


FH, FW = 2 * TH + 1, 2 * TW + 1


def back(x, fm1, skip3, w_s2, b_s2, w_s1, b_s1, w_head, alpha):
    n, h2, w2, _ = x.shape
    h, w = 2 * h2, 2 * w2
    ty, tx = (h2 + TH - 1) // TH, (w2 + TW - 1) // TW
    nblocks = n * ty * tx
    X_, F_, S_ = x.reshape(-1), fm1.reshape(-1), skip3.reshape(-1)
    ws2, ws1, wh = (np.asarray(a, np.float32).reshape(-1) for a in (w_s2, w_s1, w_head))
    pred = np.full((n * h * w * 3,), np.nan, np.float32)
    hw2 = h2 * w2
    for blk in range(nblocks):
        tile = xcd_tile(blk, nblocks)
        tx0 = (tile % tx) * TW; tile //= tx
        ty0 = (tile % ty) * TH
        f = tile // ty
        lds = np.full((FH * FW * 4,), np.nan, np.float32)
        for wave in range(4):
            a2 = []
            for c in range(3):
                c0 = 16 * c + 4 * KK
                a2.append(np.stack([np.where(c0 < 40, ws2[np.minimum(J * 40 + c0 + s4, len(ws2) - 1)], 0) for s4 in range(4)], 1))
            bs2 = np.tile(np.asarray(b_s2, np.float32)[None, :4], (64, 1))
            for mt in range(wave, NT, 4):
                t = mt * 16 + J
                live = t < HT
                hy = np.where(live, t // HW, 0); hx = np.where(live, t % HW, 0)
                gy, gx = ty0 - 1 + hy, tx0 - 1 + hx
                inside = live & (gy >= 0) & (gx >= 0) & (gy < h2) & (gx < w2)
                tex = f * hw2 + np.where(inside, gy * w2 + gx, 0)
                acc = np.zeros((64, 4), np.float32)
                for c in range(3):
                    c0 = 16 * c + 4 * KK
                    b = np.zeros((64, 4), np.float32)
                    for s4 in range(4):
                        b[:, s4] = np.where(c0 < 8, X_[np.minimum(tex * 8 + c0 + s4, len(X_) - 1)],
                                            np.where(c0 < 40, F_[np.minimum(tex * 32 + (c0 - 8) + s4, len(F_) - 1)], 0))
                        acc = mfma(a2[c][:, s4], b[:, s4], acc)
                acc = lrelu(acc + bs2, alpha)
                acc[~inside] = 0
                ly = 2 * hy + (KK >> 1) - 1; lx = 2 * hx + (KK & 1) - 1
                for l in np.nonzero(live & (ly >= 0) & (lx >= 0))[0]:
                    a0 = (ly[l] * FW + lx[l]) * 4
                    assert np.isnan(lds[a0])
                    lds[a0:a0 + 4] = acc[l]
        for tid in range(256):
            for half in range(2):
                ox, oy = tid & 31, (tid >> 5) + 8 * half
                y, xg = 2 * ty0 + oy, 2 * tx0 + ox
                if y >= h or xg >= w:
                    continue
                d = [np.float32(b_s1[o]) for o in range(4)]
                for t_ in range(4):
                    a0 = ((oy + 1 - (t_ >> 1)) * FW + ox + 1 - (t_ & 1)) * 4
                    v = lds[a0:a0 + 4]
                    assert not np.isnan(v).any()
                    for o in range(4):
                        for c in range(4):
                            d[o] = d[o] + v[c] * ws1[(t_ * 4 + o) * 4 + c]
                tex = (f * h + y) * w + xg
                p = [S_[tex * 3 + o] for o in range(3)]
                for c in range(4):
                    dv = d[c] if d[c] > 0 else np.float32(alpha) * d[c]
                    for o in range(3):
                        p[o] = p[o] + dv * wh[c * 3 + o]
                if y == 0 and xg == 0:
                    p = [0, 0, 0]
                assert np.isnan(pred[tex * 3])
                pred[tex * 3: tex * 3 + 3] = p
    return pred.reshape(n, h, w, 3)


def _oracle_intermediates(om, x, y_obs):
    """Model._call (nlt.py:141-199) keeping what the fused kernels exchange."""
    feats = {}
    stack, query_x, obs_xs = [], x, y_obs
    for i, (L, c) in enumerate(zip(om.layers, om.is_contracting)):
        if c:
            obs_ys = [O.apply_layer(L, om.wo[i], o) for o in obs_xs]
            agg = torch.stack(obs_ys, -1).mean(-1)
            obs_xs = obs_ys
            query_x = torch.cat((O.apply_layer(L, om.wq[i], query_x), agg), -1)
            stack.append(query_x)
            feats['fm%d' % i] = query_x; feats['obs%d' % i] = torch.stack(obs_ys, 1)
        else:
            if stack:
                query_x = torch.cat((query_x, stack.pop()), -1)
            feats['in%d' % i] = query_x
            query_x = O.apply_layer(L, om.wq[i], query_x)
            feats['dec%d' % i] = query_x
    return query_x, feats


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize('uv,k,add_base', [(64, 2, True), (64, 1, False)])
def test_fused_ends_emulation_matches_oracle(uv, k, add_base):
    om = O.OracleModel(depth=256, uvh=uv, uvw=uv, imh=32, imw=32, seed=5, skip_connect_base=add_base)
    batch, nn = O.synth_batch(1, uv, uv, 32, 32, 32, 32, k=k, seed=6)
    _, base, cvis, lvis = batch[:4]
    with torch.no_grad():
        x = torch.cat((base, cvis, lvis), 3)
        y_obs = [r - b for b, r in nn]
        pred, feats = _oracle_intermediates(om, x, y_obs)
        if add_base:
            pred = pred + base
        pred = T.set_left_top_corner(pred, 0)
    g = lambda t: t.detach().numpy()
    W = om.numpy_weights()
    (wq0, bq0), = W['query'][0]; (wo0, bo0), = W['obs'][0]
    (wqa, bqa), (wqb, bqb) = W['query'][1]; (woa, boa), (wob, bob) = W['obs'][1]
    (wh, bh), = W['query'][-1]
    blob = pack(dict(wq0=wq0, bq0=bq0, wo0=wo0, bo0=bo0, wqa=wqa, bqa=bqa, wqb=wqb, bqb=bqb, woa=woa, boa=boa,
                     wob=wob, bob=bob, wh=wh, bh=bh))
    nn_rgb = np.stack([g(r) for _, r in nn], 1); nn_base = np.stack([g(b) for b, _ in nn], 1)
    (wqa2, bqa2), _ = W['query'][2]; (woa2, boa2), _ = W['obs'][2]
    fm1, obs1, skip3, qtmp2, otmp2 = front(g(base), g(cvis), g(lvis), nn_rgb, nn_base, blob, add_base, 0.3,
                                           blob3=pack_l2(wqa2, bqa2, woa2, boa2))
    assert not np.isnan(fm1).any() and not np.isnan(obs1).any() and not np.isnan(skip3).any()
    with torch.no_grad():                                       # level 2's stride-2 convs (convnet.py:50-53)
        q2 = T.leaky_relu(T.conv2d_same(feats['fm1'], torch.from_numpy(wqa2), torch.from_numpy(bqa2), 2))
        o2 = torch.stack([T.leaky_relu(T.conv2d_same(feats['obs1'][:, i], torch.from_numpy(woa2), torch.from_numpy(boa2), 2))
                          for i in range(k)], 1)
    assert not np.isnan(qtmp2).any() and not np.isnan(otmp2).any()
    assert _rel(qtmp2, g(q2)) < 2e-6 and _rel(otmp2, g(o2)) < 2e-6
    assert _rel(fm1, g(feats['fm1'])) < 2e-6
    assert _rel(obs1, g(feats['obs1'])) < 2e-6
    fm0 = g(feats['fm0'])
    ref_skip = fm0 @ wh[0, 0, 4:, :] + bh + (g(base) if add_base else 0)
    assert _rel(skip3, ref_skip) < 2e-6
    n_layers = len(om.layers)
    (w_s2, b_s2), (w_s1, b_s1) = W['query'][n_layers - 2]
    x_in = g(feats['dec%d' % (n_layers - 3)])                     # previous decoder block's output (8 channels)
    assert x_in.shape[-1] == 8 and w_s2.shape == (2, 2, 4, 40)
    got = back(x_in, g(feats['fm1']), ref_skip.astype(np.float32), w_s2, b_s2, w_s1, b_s1, wh[0, 0, :4, :], 0.3)
    assert not np.isnan(got).any()
    assert _rel(got, g(pred)) < 2e-6


def test_fused_ends_emulation_partial_tiles():
    """40 x 56 UV (half-res 20 x 28: tiles of 8 x 16 do not divide it) -- layers applied directly."""
    h, w, k = 40, 56, 2
    om = O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32, seed=9)
    rng = np.random.default_rng(4)
    U = lambda *s: torch.from_numpy(rng.random(s, dtype=np.float32))
    base, cvis, lvis = U(2, h, w, 3), U(2, h, w, 1), U(2, h, w, 1)
    nn = [(U(2, h, w, 3), U(2, h, w, 3)) for _ in range(k)]
    g = lambda t: t.detach().numpy()
    W = om.numpy_weights()
    (wq0, bq0), = W['query'][0]; (wo0, bo0), = W['obs'][0]
    (wqa, bqa), (wqb, bqb) = W['query'][1]; (woa, boa), (wob, bob) = W['obs'][1]
    (wh, bh), = W['query'][-1]
    with torch.no_grad():
        o0 = [O.apply_layer(om.layers[0], om.wo[0], r - b) for b, r in nn]
        fm0 = torch.cat((O.apply_layer(om.layers[0], om.wq[0], torch.cat((base, cvis, lvis), 3)), torch.stack(o0, -1).mean(-1)), -1)
        o1 = [O.apply_layer(om.layers[1], om.wo[1], o) for o in o0]
        fm1_ref = torch.cat((O.apply_layer(om.layers[1], om.wq[1], fm0), torch.stack(o1, -1).mean(-1)), -1)
    blob = pack(dict(wq0=wq0, bq0=bq0, wo0=wo0, bo0=bo0, wqa=wqa, bqa=bqa, wqb=wqb, bqb=bqb, woa=woa, boa=boa,
                     wob=wob, bob=bob, wh=wh, bh=bh))
    (wqa2, bqa2), _ = W['query'][2]; (woa2, boa2), _ = W['obs'][2]
    fm1, obs1, skip3, qtmp2, otmp2 = front(g(base), g(cvis), g(lvis), np.stack([g(r) for _, r in nn], 1),
                                           np.stack([g(b) for b, _ in nn], 1), blob, True, 0.3, blob3=pack_l2(wqa2, bqa2, woa2, boa2))
    assert not np.isnan(fm1).any() and not np.isnan(obs1).any() and not np.isnan(skip3).any()
    with torch.no_grad():
        q2 = T.leaky_relu(T.conv2d_same(fm1_ref, torch.from_numpy(wqa2), torch.from_numpy(bqa2), 2))
        o2 = torch.stack([T.leaky_relu(T.conv2d_same(o, torch.from_numpy(woa2), torch.from_numpy(boa2), 2)) for o in o1], 1)
    assert not np.isnan(qtmp2).any() and not np.isnan(otmp2).any()
    assert _rel(qtmp2, g(q2)) < 2e-6 and _rel(otmp2, g(o2)) < 2e-6
    assert _rel(fm1, g(fm1_ref)) < 2e-6 and _rel(obs1, np.stack([g(o) for o in o1], 1)) < 2e-6
    ref_skip = (g(fm0) @ wh[0, 0, 4:, :] + bh + g(base)).astype(np.float32)
    assert _rel(skip3, ref_skip) < 2e-6
    n_layers = len(om.layers)
    Lup = om.layers[n_layers - 2]
    x = U(2, h // 2, w // 2, 8)
    with torch.no_grad():
        d = O.apply_layer(Lup, om.wq[n_layers - 2], torch.cat((x, fm1_ref), -1))
        pred = torch.from_numpy(g(d) @ wh[0, 0, :4, :] + ref_skip)
        pred = T.set_left_top_corner(pred, 0)
    (w_s2, b_s2), (w_s1, b_s1) = W['query'][n_layers - 2]
    got = back(g(x), g(fm1_ref), ref_skip, w_s2, b_s2, w_s1, b_s1, wh[0, 0, :4, :], 0.3)
    assert not np.isnan(got).any() and _rel(got, g(pred)) < 2e-6


def test_front_lds_patterns_are_bank_conflict_free():
    """ds_read/write_b128 serves 16 lanes per pass (16 lanes x 16 B = all 64 banks): every 16-lane group of the
    front kernel's LDS accesses must touch 16 different 16-byte slots modulo 16 (fused.hip stages 1-3)."""
    def free(addr_floats):
        slots = (np.asarray(addr_floats) // 4).reshape(4, 16) % 16
        return all(len(set(g)) == 16 for g in slots)
    for mt in range(NT - 1):                                           # stage-1 writes (the last tile is ragged)
        assert free((KK * PLT + mt * 16 + J) * 4)
    for r in range(8):                                                 # stage-2 reads, both rows of a wave
        for t_ in range(4):
            assert free((KK * PLT + (r + (t_ >> 1)) * HW + J + (t_ & 1)) * 4)
    for r in range(8):                                                 # stage-3 writes
        par = J & 1
        assert free((KK * 128 + par * 64 + ((r * 8 + (J >> 1)) ^ ((((r >> 1) & 1) ^ par) * 8))) * 4)
    for ct in range(2):                                                # stage-3 reads
        t2 = ct * 16 + J
        Y, X = t2 >> 3, t2 & 7
        src = ((KK & 1) * 64 + (((2 * Y + (KK >> 1)) * 8 + X) ^ (((Y & 1) ^ (KK & 1)) * 8))) * 4
        for c4 in range(4):
            assert free(src + 512 * c4)
