"""CPU: host / index logic that is new in round 6.

  * networks/elements.PackRegistry after the advisor's r05 findings: a buffer the census retired is re-activated and re-packed IN
    PLACE whenever it is asked for again (whatever its version stamp says); the keys a recorded launch tape reads are kept with
    the tape and re-touched on replay, so a census started meanwhile cannot retire them;
  * csrc/front_ovr.hip's strip schedule: a wave's interior strips (pipelined loop, `next_interior`) and border strips (second
    loop) together are exactly its strips, each once; every prefetch index of the pipelined loop is a valid interior strip."""
import numpy as np
import pytest
import torch

from nlt_amd.networks.elements import Conv2D, PackRegistry


def _layer(reg):
    c = Conv2D(8, 2, 1)
    c.kernel, c.bias, c.cin, c.built = torch.ones(2, 2, 4, 8), torch.zeros(8), 4, True
    c._registry = reg
    return c


def test_a_retired_packed_buffer_is_reactivated_and_repacked_in_place(monkeypatch):
    from nlt_amd import _capi as C
    monkeypatch.setattr(C, 'repack_table', lambda rows, dev: (None, len(rows), 0))
    monkeypatch.setattr(C, 'repack_weights', lambda *a: None)
    reg = PackRegistry()
    c = _layer(reg)
    make = lambda: torch.tensor([float(c.kernel.sum())])
    desc = dict(kind=0, mode=c.mode, c0=4, c1=0, cout=8, tn=0, lo=0, full=8)
    buf = c._cached_pack('k', make, desc)
    assert float(buf) == 128.0 and reg.owns(c, 'k')
    reg.begin_census(passes=1)
    reg.tick(); reg.tick()                                   # nothing asked for the buffer while the census ran: retired
    assert reg.is_inactive(c, 'k')
    # (a) same version, inactive: the fast path must not hand it out inactive
    again = c._cached_pack('k', make, desc)
    assert again is buf and not reg.is_inactive(c, 'k')
    # (b) retired AND the weights were rewritten meanwhile: re-packed in place, same tensor object (tapes keep its address)
    reg.begin_census(passes=1)
    reg.tick(); reg.tick()
    assert reg.is_inactive(c, 'k')
    c.kernel.add_(1.0)
    again = c._cached_pack('k', make, desc)
    assert again is buf and float(buf) == 256.0 and not reg.is_inactive(c, 'k')


def test_keys_read_by_a_recorded_tape_survive_a_census_started_by_somebody_else():
    reg = PackRegistry()
    a, b = _layer(reg), _layer(reg)
    desc = dict(kind=0, mode=a.mode, c0=4, c1=0, cout=8, tn=0, lo=0, full=8)
    reg.begin_record()                                       # a plan records a launch tape ...
    a._cached_pack('ka', lambda: torch.zeros(1), desc)
    keys = reg.end_record()
    b._cached_pack('kb', lambda: torch.zeros(1), desc)       # (asked for outside any recording, by another plan's trial)
    assert keys == frozenset({(id(a), 'ka')})
    reg.begin_census(passes=2)                               # ... another plan re-tunes and starts a census
    reg.tick(); reg.touch_keys(keys)                         # the first plan REPLAYS its tape: no _cached_pack call, but it re-touches
    reg.tick(); reg.touch_keys(keys)
    reg.tick()                                               # census over
    assert not reg.is_inactive(a, 'ka') and reg.is_inactive(b, 'kb')
    assert reg.end_record() == frozenset()                   # (no recording open: nothing collected)


def _wave_tiles(ntiles, grid, nw=8):
    per = (ntiles + 7) >> 3
    stride = (grid >> 3) * nw
    for blk in range(grid):
        t_lo = (blk & 7) * per
        t_hi = min(t_lo + per, ntiles)
        for wv in range(nw):
            first = t_lo + wv * (grid >> 3) + (blk >> 3)
            yield first, t_hi, stride


@pytest.mark.parametrize('h,w,n', [(64, 64, 2), (72, 40, 1), (128, 96, 3), (36, 132, 2), (1024, 1024, 1), (8, 8, 1)])
def test_front_ovr_strip_schedule_covers_every_strip_once(h, w, n):
    SH, SW, AH, AW = 4, 16, 5, 17
    h2, w2 = h // 2, w // 2
    ty, tx = (h2 + SH - 1) // SH, (w2 + SW - 1) // SW
    ntiles = n * ty * tx
    groups = min((ntiles + 7) // 8, 32)
    grid = 8 * groups

    def interior(t):
        tx0 = (t % tx) * SW
        ty0 = ((t // tx) % ty) * SH
        return ty0 + AH <= h2 and tx0 + AW <= w2
    seen = np.zeros(ntiles, int)
    for first, t_hi, stride in _wave_tiles(ntiles, grid):
        if first >= t_hi:
            continue

        def next_interior(t):
            t += stride
            while t < t_hi and not interior(t):
                t += stride
            return t
        cur = first if interior(first) else next_interior(first)
        fast = []
        if cur < t_hi:
            nx = next_interior(cur)
            while True:
                fast.append(cur)
                pf = nx if nx < t_hi else cur                # maps + raw rows staged for the next strip: clamped, always valid
                n2 = next_interior(pf)
                pf2 = n2 if n2 < t_hi else pf
                assert pf < t_hi and interior(pf) and pf2 < t_hi and interior(pf2)
                if nx >= t_hi:
                    break
                cur, nx = nx, next_interior(nx)
        slow = [t for t in range(first, t_hi, stride) if not interior(t)]
        assert all(interior(t) for t in fast) and sorted(fast + slow) == list(range(first, t_hi, stride))
        for t in fast + slow:
            seen[t] += 1
    assert (seen == 1).all()
