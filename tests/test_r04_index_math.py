"""CPU: lane- and wave-level NumPy emulation of the index math that is NEW in round 4's kernels (the arithmetic on the values is
the MFMA's; what can go wrong is WHICH value meets which):

  * the persistent tile schedules of front4 / front5 (8-wave workgroups) and conv_c32 (256-thread workgroups): every tile is taken
    by exactly one wave / workgroup, XCD x = blockIdx & 7 owns a contiguous run, ragged tile counts and small grids included;
  * the staged-item sequence of a persistent front wave (observation 0 .. k - 1, query inputs, next strip's observation 0 ...):
    every item is stored exactly once, after it was loaded, before the stage that reads it, and never over data still to be read;
  * front5's register hand-offs: lane (kk, j = (Y, X)) of stage-1 tile (u, v) / stage-2 tile (a, b) holds exactly the texel the
    next stage's tap needs -- checked by running the whole three-conv chain of one strip through the kernel's dataflow in float64
    and comparing it with the three convs applied to the image;
  * dec_block10's work units: (column tile, row half) pairs cover the haloed tile's 10 column tiles x MT row tiles once, and the
    chunk -> (x | skip) channel mapping walks the virtual concat in order.
"""
import numpy as np
import pytest

LANE = np.arange(64)
KK, J = LANE >> 4, LANE & 15


# ------------------------------------------------------------------------------------------------ persistent schedules
def front_schedule(ntiles, grid, nwaves=8):
    """(block, wave) -> list of tiles, as csrc/front4.hip / front5.hip compute it."""
    per = (ntiles + 7) >> 3
    stride = (grid >> 3) * nwaves
    out = {}
    for b in range(grid):
        t_lo = (b & 7) * per
        t_hi = min(t_lo + per, ntiles)
        for wv in range(nwaves):
            tile, seq = t_lo + wv * (grid >> 3) + (b >> 3), []
            while tile < t_hi:
                seq.append(tile)
                tile += stride
            out[(b, wv)] = seq
    return out


def front_grid(tiles, nwaves=8):
    per_xcd = (tiles + 7) // 8
    return 8 * min(32, per_xcd)


@pytest.mark.parametrize('tiles', [1, 2, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1000, 2048, 2049, 16384, 16384 + 13, 4 * 64 * 17])
def test_front_schedule_takes_every_strip_once_and_keeps_an_xcd_contiguous(tiles):
    grid = front_grid(tiles)
    assert grid % 8 == 0 and 8 <= grid <= 256
    sched = front_schedule(tiles, grid)
    seen = np.zeros(tiles, np.int64)
    per = (tiles + 7) >> 3
    for (b, wv), seq in sched.items():
        for t in seq:
            seen[t] += 1
            assert t // per == (b & 7)                                          # the XCD's own run of tiles
    assert (seen == 1).all()
    # balance: waves of one XCD differ by at most one strip
    for x in range(8):
        counts = [len(seq) for (b, wv), seq in sched.items() if (b & 7) == x]
        assert max(counts) - min(counts) <= 1
    # concurrently resident waves of an XCD work on one contiguous block of strips per round
    for x in range(8):
        rounds = {}
        for (b, wv), seq in sched.items():
            if (b & 7) == x:
                for r, t in enumerate(seq):
                    rounds.setdefault(r, []).append(t)
        for r, ts in rounds.items():
            ts = sorted(ts)
            assert ts == list(range(ts[0], ts[0] + len(ts)))
    # a small input still spreads over the CUs: no workgroup holds two strips while another of its XCD holds none
    for x in range(8):
        per_wg = {}
        for (b, wv), seq in sched.items():
            if (b & 7) == x:
                per_wg[b] = per_wg.get(b, 0) + len(seq)
        assert max(per_wg.values()) - min(per_wg.values()) <= 8
        if sum(per_wg.values()) >= len(per_wg):                              # the XCD has at least one strip per workgroup
            assert min(per_wg.values()) >= 1


def c32_schedule(ntiles, grid):
    per = (ntiles + 7) >> 3
    stride = grid >> 3
    out = {}
    for b in range(grid):
        t_hi = min(((b & 7) + 1) * per, ntiles)
        tile, seq = (b & 7) * per + (b >> 3), []
        while tile < t_hi:
            seq.append(tile)
            tile += stride
        out[b] = seq
    return out


@pytest.mark.parametrize('tiles', [1, 5, 8, 9, 100, 511, 512, 513, 2048, 2048 + 3, 8192])
def test_conv_c32_schedule_takes_every_tile_once(tiles):
    grid = 8 * min(64, (tiles + 7) // 8)
    sched = c32_schedule(tiles, grid)
    seen = np.zeros(tiles, np.int64)
    for b, seq in sched.items():
        for t in seq:
            seen[t] += 1
    assert (seen == 1).all()


# ------------------------------------------------------------------------------------------------ staged-item pipeline
@pytest.mark.parametrize('k', [1, 2, 3, 4, 7])
@pytest.mark.parametrize('nstrips', [1, 2, 5])
def test_staged_item_sequence_of_a_persistent_front_wave(k, nstrips):
    """Event-level replay of the loop in front4_kernel / front5_kernel: `st` = the item in flight in registers, `lds_o` = what the raw
    observation tile holds, `lds_q` = the query tiles (front4 keeps them apart; for front5, where they share the observation's tile,
    the same replay with one buffer is run below)."""
    for shared in (False, True):
        st, lds_o, lds_q = None, None, None
        log = []

        def load(item):
            nonlocal st
            assert st is None, "the staging registers still hold an item"
            st = item

        def store():
            nonlocal st, lds_o, lds_q
            assert st is not None
            if st[1] == 'Q' and not shared:
                lds_q = st
            else:
                lds_o = st
                if st[1] == 'Q':
                    lds_q = st
            st = None

        # prologue
        load((0, 0)); store()
        load((0, 1) if k > 1 else (0, 'Q'))
        for s in range(nstrips):
            has_next = s + 1 < nstrips
            for i in range(k):
                assert lds_o == (s, i), (lds_o, s, i)                        # stage 1 reads observation i of strip s
                log.append((s, i))
                store()                                                      # item i + 1 (or the query inputs)
                if i + 2 < k:
                    load((s, i + 2))
                elif i + 2 == k:
                    load((s, 'Q'))
                elif has_next:
                    load((s + 1, 0))
            assert lds_q == (s, 'Q')                                         # the query path's stage 1
            log.append((s, 'Q'))
            if has_next:
                store()
                load((s + 1, 1) if k > 1 else (s + 1, 'Q'))
        assert st is None
        assert log == [(s, x) for s in range(nstrips) for x in list(range(k)) + ['Q']]


# ------------------------------------------------------------------------------------------------ front5 register hand-offs
def lrelu(v, a=0.3):
    return np.where(v > 0, v, a * v)


def conv_same(x, w, stride):
    """TF Conv2D 'same' for k = 2: pads bottom / right.  x [H, W, Cin], w [2, 2, Cin, Cout]."""
    H, W, _ = x.shape
    xp = np.pad(x, ((0, 1), (0, 1), (0, 0)))
    oh, ow = (H + stride - 1) // stride, (W + stride - 1) // stride
    out = np.zeros((oh, ow, w.shape[3]))
    for a in range(2):
        for b in range(2):
            out += np.einsum('hwc,co->hwo', xp[a:a + H:stride, b:b + W:stride][:oh, :ow], w[a, b])
    return out


@pytest.mark.parametrize('h,w', [(16, 64), (8, 32), (12, 40), (4, 4), (20, 36)])
def test_front5_tile_dataflow_is_the_three_conv_chain(h, w):
    """raw [h, w, 3] -> stride-2 conv (3 -> 16, stands for the folded L0 + L1.s2) + lrelu -> stride-1 conv (16 -> 16) + lrelu ->
    stride-2 conv (16 -> 32) + lrelu, evaluated (a) directly and (b) strip by strip with csrc/front5.hip's tiles: stage 1 produces
    T[u][v] at level-1 texel (2Y + u, 2X + v) from raw texel (4Y + 2u + tap / 2, 4X + 2v + tap % 2), zeroed outside the image;
    stage 2 produces O[a][b] = sum over taps (a', b') of W1[a', b'] . T[a + a'][b + b']; stage 3 sums W3[a, b] . O[a][b]."""
    rng = np.random.default_rng(h * 100 + w)
    raw = rng.standard_normal((h, w, 3))
    w_s2, w_s1, w_l2 = rng.standard_normal((2, 2, 3, 16)), rng.standard_normal((2, 2, 16, 16)), rng.standard_normal((2, 2, 16, 32))
    t1 = lrelu(conv_same(raw, w_s2, 2))
    o1 = lrelu(conv_same(t1, w_s1, 1))
    ref = lrelu(conv_same(o1, w_l2, 2))
    h2, w2, h4, w4 = h // 2, w // 2, h // 4, w // 4
    got = np.full((h4, w4, 32), np.nan)
    got_o1 = np.full((h2, w2, 16), np.nan)
    SH, SW = 4, 16
    for ty0 in range(0, h2, SH):
        for tx0 in range(0, w2, SW):
            for j in range(16):                                              # one lane column = one level-2 texel (Y, X)
                Y, X = j >> 3, j & 7
                lim_r, lim_c = h2 - ty0 - 2 * Y, w2 - tx0 - 2 * X
                T = np.zeros((3, 3, 16))
                for u in range(3):
                    for v in range(3):
                        acc = np.zeros(16)
                        for tap in range(4):                                 # lane group kk = tap of the stride-2 conv
                            ry, rx = 2 * ty0 + 4 * Y + 2 * u + (tap >> 1), 2 * tx0 + 4 * X + 2 * v + (tap & 1)
                            if u < lim_r and v < lim_c:                      # (outside: whatever the staged tile holds; masked below)
                                assert ry < h and rx < w
                                acc += raw[ry, rx] @ w_s2[tap >> 1, tap & 1]
                        T[u, v] = lrelu(acc) if (u < lim_r and v < lim_c) else 0.0
                O = np.zeros((2, 2, 16))
                for a in range(2):
                    for b in range(2):
                        for a1 in range(2):                                  # K block = tap row a1: tiles (a + a1, b) | (a + a1, b + 1)
                            for b1 in range(2):
                                O[a, b] += T[a + a1, b + b1] @ w_s1[a1, b1]
                O = lrelu(O)
                for a in range(2):
                    for b in range(2):
                        if a < lim_r and b < lim_c:
                            assert np.isnan(got_o1[ty0 + 2 * Y + a, tx0 + 2 * X + b]).all(), "two owners of a level-1 texel"
                            got_o1[ty0 + 2 * Y + a, tx0 + 2 * X + b] = O[a, b]
                gy2, gx2 = (ty0 >> 1) + Y, (tx0 >> 1) + X
                if gy2 < h4 and gx2 < w4:
                    acc3 = np.zeros(32)
                    for a in range(2):                                       # K block = tap row a: tiles (a, 0) | (a, 1)
                        for b in range(2):
                            acc3 += O[a, b] @ w_l2[a, b]
                    assert np.isnan(got[gy2, gx2]).all()
                    got[gy2, gx2] = lrelu(acc3)
    assert not np.isnan(got).any() and not np.isnan(got_o1).any()
    np.testing.assert_allclose(got_o1, o1, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_front5_k_block_slots_pair_the_tiles_with_the_weight_fragments():
    """A bf16 K block of 32: lane group kk, slot e < 4 = channel 4 kk + e of the FIRST tap's tile, e >= 4 = channel 4 kk + e - 4 of the
    second's.  The weight fragment (a1_load: tap = 2 a' + eh, channels 4 kk ..) must use the same (tap, channel) per slot."""
    for a1 in range(2):
        for kk in range(4):
            for e in range(8):
                tile_tap = (a1, e >> 2)                                      # pair8(T[.][b], T[.][b + 1])
                tile_ch = 4 * kk + (e & 3)
                frag_tap = 2 * a1 + (e >> 2)                                 # a1_load: r[2 a + eh], eh = e >> 2
                frag_ch = 4 * kk + (e & 3)                                   # blob [tap][lane' = kk * 16 + o][s4]
                assert (frag_tap >> 1, frag_tap & 1) == tile_tap and frag_ch == tile_ch
    # the raw staging offsets of tile (u, v): immediates on one lane base, inside the 10 x 34 staged tile
    R3, R1 = 104, 40
    Y, X = J >> 3, J & 7
    rd3 = (4 * Y + (KK >> 1)) * R3 + (4 * X + (KK & 1)) * 3
    rd1 = (4 * Y + (KK >> 1)) * R1 + 4 * X + (KK & 1)
    for u in range(3):
        for v in range(3):
            o3 = rd3 + u * 2 * R3 + v * 6
            row, col = o3 // R3, (o3 % R3) // 3
            np.testing.assert_array_equal(row, 4 * Y + 2 * u + (KK >> 1))
            np.testing.assert_array_equal(col, 4 * X + 2 * v + (KK & 1))
            assert row.max() <= 9 and col.max() <= 33
            o1 = rd1 + u * 2 * R1 + v * 2
            np.testing.assert_array_equal(o1 // R1, row)
            np.testing.assert_array_equal(o1 % R1, col)


# ------------------------------------------------------------------------------------------------ dec_block10
@pytest.mark.parametrize('C', [8, 16])
def test_dec_block10_units_and_chunks(C):
    NT, MT = 10, C // 4
    MH, NI, K = MT // 2, NT // 2, 10 * C
    seen = np.zeros((NT, MT), np.int64)
    for wave in range(4):
        mh, cg = wave & 1, wave >> 1
        for i in range(NI):
            for m in range(MH):
                seen[cg + 2 * i, mh * MH + m] += 1
    assert (seen == 1).all()
    # chunk ch, lane group kk -> concat channels 16 ch + 4 kk .. + 3; x holds the first 2C, skip (8C) the rest
    chans = []
    for ch in range(K // 16):
        for kk in range(4):
            c0 = 16 * ch + 4 * kk
            src = 'x' if c0 < 2 * C else 'skip'
            off = c0 if src == 'x' else c0 - 2 * C
            assert off % 4 == 0 and off + 4 <= (2 * C if src == 'x' else 8 * C)
            chans.append((src, off))
    assert chans == [('x', c) for c in range(0, 2 * C, 4)] + [('skip', c) for c in range(0, 8 * C, 4)]
    # this lane's four output columns of row tile m: col = 16 m + 4 kk = (ab, o0 ..): o0 independent of m, ab covers the 4 sub-texels
    for kk in range(4):
        assert len({(16 * m + 4 * kk) % C for m in range(MT)}) == 1
    assert sorted({(16 * m + 4 * kk) // C for m in range(MT) for kk in range(4)}) == [0, 1, 2, 3]
