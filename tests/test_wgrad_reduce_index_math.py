"""CPU emulation of the thread -> (slice, entry) mapping of the second passes of the weight-gradient kernels
(csrc/wgrad_tile.hip wgrad_reduce_kernel<MODE, false>, csrc/wgrad_narrow.hip wgrad_narrow_reduce_kernel; 16-byte loads, 8 in
flight per thread since r03_d): every (slice, entry) pair must be added exactly once, whatever the number of slices (the
unrolled-by-8 loop + its tail) and whatever the padded block size (the narrow kernel's last workgroup is ragged)."""
import numpy as np
import pytest


def tile_reduce(ws, msplits, per_slice):
    """wgrad_reduce_kernel<MODE, false>: workgroup = 64 entries = 16 quads x 16 slice groups."""
    assert per_slice % 64 == 0
    out = np.zeros(per_slice, np.float32)
    count = np.zeros((msplits, per_slice), np.int32)
    for blk in range(per_slice // 64):
        part = np.zeros((16, 64), np.float32)
        for t in range(256):
            q, g = t & 15, t >> 4
            idx = blk * 64 + 4 * q
            s0, s1 = np.zeros(4, np.float32), np.zeros(4, np.float32)
            ms = g
            while ms + 112 < msplits:
                v = [ws[ms + 16 * j, idx:idx + 4] for j in range(8)]
                for j in range(8):
                    count[ms + 16 * j, idx:idx + 4] += 1
                s0 = s0 + ((v[0] + v[1]) + (v[2] + v[3]))
                s1 = s1 + ((v[4] + v[5]) + (v[6] + v[7]))
                ms += 128
            while ms < msplits:
                s0 = s0 + ws[ms, idx:idx + 4]
                count[ms, idx:idx + 4] += 1
                ms += 16
            part[g, 4 * q:4 * q + 4] = s0 + s1
        for e in range(64):
            t4 = [(part[4 * a, e] + part[4 * a + 1, e]) + (part[4 * a + 2, e] + part[4 * a + 3, e]) for a in range(4)]
            out[blk * 64 + e] = (t4[0] + t4[1]) + (t4[2] + t4[3])
    return out, count


def narrow_reduce(ws, msplits, per):
    """wgrad_narrow_reduce_kernel: workgroup = 32 entries = 8 quads x 32 slice groups; PER is a multiple of 16 only."""
    assert per % 16 == 0
    out = np.full(per, np.nan, np.float32)
    count = np.zeros((msplits, per), np.int32)
    flat = ws.reshape(-1)
    for blk in range((per + 31) // 32):
        part = np.zeros((32, 32), np.float32)
        for t in range(256):
            q, g = t & 7, t >> 3
            base = blk * 32 + 4 * q
            live = base < per
            b = base if live else 0                                  # (a dead quad re-reads the block start; its sum is dropped)
            s0, s1 = np.zeros(4, np.float32), np.zeros(4, np.float32)
            ms = g
            while ms + 224 < msplits:
                v = [flat[(ms + 32 * j) * per + b:(ms + 32 * j) * per + b + 4] for j in range(8)]
                if live:
                    for j in range(8):
                        count[ms + 32 * j, b:b + 4] += 1
                s0 = s0 + ((v[0] + v[1]) + (v[2] + v[3]))
                s1 = s1 + ((v[4] + v[5]) + (v[6] + v[7]))
                ms += 256
            while ms < msplits:
                s0 = s0 + flat[ms * per + b:ms * per + b + 4]
                if live:
                    count[ms, b:b + 4] += 1
                ms += 32
            part[g, 4 * q:4 * q + 4] = s0 + s1
        for e in range(32):
            idx = blk * 32 + e
            if idx >= per:
                continue
            t4 = [sum(part[8 * a + bb, e] for bb in range(8)) for a in range(4)]
            out[idx] = (t4[0] + t4[1]) + (t4[2] + t4[3])
    return out, count


@pytest.mark.parametrize('msplits', [1, 5, 16, 17, 112, 113, 128, 129, 300])
@pytest.mark.parametrize('per_slice', [64, 192])
def test_tile_reduce_adds_every_slice_entry_once(msplits, per_slice):
    rng = np.random.default_rng(msplits + per_slice)
    ws = rng.standard_normal((msplits, per_slice)).astype(np.float32)
    out, count = tile_reduce(ws, msplits, per_slice)
    assert (count == 1).all()
    ref = ws.astype(np.float64).sum(0)
    assert np.abs(out - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()) * np.sqrt(msplits)


@pytest.mark.parametrize('msplits', [1, 7, 32, 33, 224, 225, 256, 257, 600])
@pytest.mark.parametrize('per', [16 * 17, 16 * 33, 32 * 129, 32 * 65])       # NC * (MT * 16 + 1): 16 x odd is not a multiple of 32
def test_narrow_reduce_adds_every_slice_entry_once(msplits, per):
    if msplits > 300 and per > 2000:
        pytest.skip("covered by the smaller blocks")
    rng = np.random.default_rng(msplits + per)
    ws = rng.standard_normal((msplits + 1, per)).astype(np.float32)       # (+ 1: the slice-sum row the k2s2 bias pass uses)
    out, count = narrow_reduce(ws, msplits, per)
    assert (count[:msplits] == 1).all() and not np.isnan(out).any()
    ref = ws[:msplits].astype(np.float64).sum(0)
    assert np.abs(out - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()) * np.sqrt(msplits)
