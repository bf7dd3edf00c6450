"""CPU: lane- and wave-level NumPy emulation of the index math that is NEW in round 4's kernels (the arithmetic on the values is
the MFMA's; what can go wrong is WHICH value meets which):

  * the persistent tile schedules of front4 / front_ovr (8-wave workgroups) and conv_c32 (256-thread workgroups): every tile is taken
    by exactly one wave / workgroup, XCD x = blockIdx & 7 owns a contiguous run, ragged tile counts and small grids included;
  * the staged-item sequence of a persistent front wave (observation 0 .. k - 1, query inputs, next strip's observation 0 ...):
    every item is stored exactly once, after it was loaded, before the stage that reads it, and never over data still to be read;
  * dec_block10's work units: (column tile, row half) pairs cover the haloed tile's 10 column tiles x MT row tiles once, and the
    chunk -> (x | skip) channel mapping walks the virtual concat in order.
"""
import numpy as np
import pytest

LANE = np.arange(64)
KK, J = LANE >> 4, LANE & 15


# ------------------------------------------------------------------------------------------------ persistent schedules
def front_schedule(ntiles, grid, nwaves=8):
    """(block, wave) -> list of tiles, as csrc/front4.hip / front_ovr.hip compute it."""
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
    """Event-level replay of the loop in front4_kernel: `st` = the item in flight in registers, `lds_o` = what the raw
    observation tile holds, `lds_q` = the query tiles (front4 keeps them apart; the same replay with ONE shared buffer is run below as well)."""
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
