"""CPU: statement-by-statement NumPy emulation of the data movement that is NEW in csrc/front4.hip (everything after the
raw values reach the MFMA operands is front_kernel's arithmetic, emulated in tests/test_fused_index_math.py):

  * staging: lane -> (raw row, piece) items of a wave's 10 x 34-texel raw strip, natural row layout in LDS
    (offset = item * piece floats), validity / alignment of every piece that is loaded for real;
  * stage 1: lane -> raw-tile offsets (rd3, the 1-channel offset), i.e. every lane must read exactly the texel
    front_kernel gathers from global memory: x[f, 2*gy + (kk >> 1), 2*gx + (kk & 1), :] for inside texels;
  * the stage-1 tile -> stage-2 tap addressing, and the level-1 tile of stage 3 (x-parity planes, xor swizzle): every
    read finds what was written for that texel, and the ds_read_b128 / ds_read_b32 accesses are bank-conflict-free;
  * the uint8 path: q = u * r; q += fma(-255, q, u) * r  ==  float32(float64(u) / 255) for all 256 bytes
    (nlt/datasets/nlt.py:131-136: xm.img.normalize_uint then astype(float32)).
"""
import numpy as np
import pytest

SH, SW = 4, 16
AH, AW = SH + 1, SW + 1
AT = AH * AW
NC = (AT + 15) // 16
SLOTS = NC * 16
XH = 2 * AH
R3, R1 = 104, 40
LANE = np.arange(64)
KK, J = LANE >> 4, LANE & 15
# lanes serviced together by one LDS cycle (MI355X_MICROARCH.md, LDS section)
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def test_u8_unit_is_exact_for_every_byte():
    u = np.arange(256)
    ref = (u.astype(np.float64) / 255.0).astype(np.float32)
    r = np.float32(1) / np.float32(255)
    uf = u.astype(np.float32)
    q = (uf * r).astype(np.float32)                                           # __fmul_rn
    # fma emulated exactly in float64: |255 q| < 2^8 with 24 significant bits, u an integer < 2^8 -> the sum is exact
    e = (uf.astype(np.float64) - 255.0 * q.astype(np.float64)).astype(np.float32)
    got = (q.astype(np.float64) + e.astype(np.float64) * np.float64(r)).astype(np.float32)
    np.testing.assert_array_equal(got, ref)


def stage(img, f, ty0, tx0, u8, channels):
    """load3 + store3 (channels = 3) or load1 + store1 (channels = 1) of one array: the wave's raw LDS tile."""
    h, w = img.shape[1:3]
    flat = img.reshape(img.shape[0], -1)
    if channels == 3:
        N, P, E, row = (13, 3, 8, R3) if u8 else (26, 5, 4, R3)
    else:
        N, P, E, row = (5, 1, 8, R1) if u8 else (9, 2, 4, R1)
    raw = np.full(XH * row, np.nan)
    for lane in range(64):
        for p in range(P):
            item = p * 64 + lane
            r, i = divmod(item, N)
            gy = 2 * ty0 + r
            ok = item < XH * N and gy < h and channels * (2 * tx0) + E * i < channels * w
            off = (gy * w + 2 * tx0) * channels + E * i if ok else 0
            assert off % E == 0 and off + E <= h * w * channels               # aligned, inside the frame (never faults)
            if ok:
                assert (off + E - 1) // (channels * w) == gy                   # a real piece lies inside its own image row
            if item >= XH * N:
                continue
            piece = flat[f, off:off + E]                                      # unconditional load (frame start when not ok)
            dst = item * E if channels == 3 else r * R1 + E * i
            assert np.isnan(raw[dst:dst + E]).all(), "two writers"
            raw[dst:dst + E] = piece
    return raw


@pytest.mark.parametrize('u8', [False, True])
@pytest.mark.parametrize('h,w', [(64, 96), (40, 72), (32, 32), (8, 8)])
def test_staging_then_stage1_reads_the_texels_front_kernel_gathers(u8, h, w):
    if u8 and w % 8:
        pytest.skip("uint8 pieces need w % 8 == 0")
    rng = np.random.default_rng(h + w)
    img3 = rng.integers(1, 255, size=(2, h, w, 3)).astype(np.float64)
    img1 = rng.integers(1, 255, size=(2, h, w, 1)).astype(np.float64)
    h2, w2 = h // 2, w // 2
    for f in range(2):
        for ty0 in range(0, h2, SH):
            for tx0 in range(0, w2, SW):
                raw3 = stage(img3, f, ty0, tx0, u8, 3)
                raw1 = stage(img1, f, ty0, tx0, u8, 1)
                for c in range(NC):
                    t = c * 16 + J
                    live = t < AT
                    hy = np.where(live, t // AW, 0); hx = np.where(live, t % AW, 0)
                    inside = live & (ty0 + hy < h2) & (tx0 + hx < w2)
                    rd3 = (2 * hy + (KK >> 1)) * R3 + (2 * hx + (KK & 1)) * 3
                    rd1 = (2 * hy + (KK >> 1)) * R1 + 2 * hx + (KK & 1)
                    for l in range(64):
                        if inside[l]:                                         # (outside texels: any finite value, masked later)
                            y, x = 2 * (ty0 + hy[l]) + (KK[l] >> 1), 2 * (tx0 + hx[l]) + (KK[l] & 1)
                            np.testing.assert_array_equal(raw3[rd3[l]: rd3[l] + 3], img3[f, y, x])
                            assert raw1[rd1[l]] == img1[f, y, x, 0]
                        else:
                            assert not np.isnan(raw3[rd3[l]: rd3[l] + 3]).any() and not np.isnan(raw1[rd1[l]])
                    # ds_read_b32, two groups of 32 lanes, 32 banks: distinct addresses on one bank must not exceed 2
                    for c3 in range(3):
                        for g in (range(0, 32), range(32, 64)):
                            a = np.unique((rd3 + c3)[list(g)])
                            assert max(np.bincount(a % 32)) <= 2


def test_stage2_taps_find_the_stage1_tile_and_level1_tile_round_trips():
    # stage-1 tile: lane (kk = channel quad, j) writes slot c*16 + j of plane kk; stage 2 reads slot (r + a) * AW + j + b
    tile = np.full((4, SLOTS), -1, np.int64)
    for c in range(NC):
        for l in range(64):
            tile[KK[l], c * 16 + J[l]] = c * 16 + J[l]                        # value = haloed texel index t
    for t4 in range(4):
        a, b = t4 >> 1, t4 & 1
        for r in range(SH):
            slot = (r + a) * AW + J + b
            assert slot.max() < AT
            np.testing.assert_array_equal(tile[KK, slot], (r + a) * AW + J + b)
            for g in B128_GROUPS:                                             # ds_read_b128: 16 different 16-byte slots
                assert len(set(((KK[g] * SLOTS + slot[g]) % 16).tolist())) == 16
    # level-1 tile: lane (kk = channel quad, j = x) writes row r; stage 3 lane (kk2 = tap, j2 = (Y, X)) reads channel quad c4
    l1 = np.full(4 * 64, -1, np.int64)
    for r in range(SH):
        wr = KK * 64 + (J & 1) * 32 + (((r * 8) + (J >> 1)) ^ ((J & 1) * 8))
        assert (l1[wr] == -1).all()
        l1[wr] = KK * 1000 + r * 16 + J                                       # (quad, row, x)
    assert (l1 >= 0).all()
    Y, X = J >> 3, J & 7
    rd = (KK & 1) * 32 + ((((2 * Y + (KK >> 1)) * 8) + X) ^ ((KK & 1) * 8))
    for c4 in range(4):
        got = l1[c4 * 64 + rd]
        want = c4 * 1000 + (2 * Y + (KK >> 1)) * 16 + 2 * X + (KK & 1)        # level-1 texel (2Y + a, 2X + b), quad c4
        np.testing.assert_array_equal(got, want)
        for g in B128_GROUPS:
            assert len(set((rd[g] % 16).tolist())) == 16
