"""CPU: NumPy emulation of the STAGING that is new in csrc/train_fused.hip's front_bwd2_kernel: the lane -> (array, raw row, 16-byte
unit) descriptors of a wave's run of 16 half-resolution texels, the wave-private LDS map, and the 4-byte operand reads that follow.
Every raw element the first-generation kernel gathers from global memory must be found at the LDS address the new kernel reads,
every 16-byte load must be aligned and inside its tensor, and every LDS float of the map must be written exactly once."""
import numpy as np
import pytest

FB_QB, FB_QC, FB_QL, FB_QD, FB_DYQ, FB_OBS = 0, 192, 256, 320, 512, 768
FB_O_RGB, FB_O_BASE, FB_O_DY, FB_O_SIZE = 0, 192, 384, 640


def stage_run(arrs, K, n, h, w, run):
    """issue() + park(): returns the wave's LDS image (NaN = never written) for run index `run`."""
    base, cvis, lvis, dpred, nn_rgb, nn_base, dy1q, dy1o = arrs
    h2, w2 = h // 2, w // 2
    rpr = w2 // 16
    hw, hw2 = h * w, h2 * w2
    xq = run % rpr
    row = run // rpr
    y, f = row % h2, row // h2
    L = np.full(FB_OBS + K * FB_O_SIZE, np.nan)
    written = np.zeros(L.size, int)

    def put(dst, src_flat, off):
        assert off % 4 == 0 and 0 <= off and off + 4 <= src_flat.size, "16-byte load: aligned and inside the tensor"
        L[dst:dst + 4] = src_flat[off:off + 4]
        written[dst:dst + 4] += 1

    t0 = f * hw + 2 * y * w + 32 * xq
    for lane in range(64):
        # slot 0
        if lane < 48:
            r0 = int(lane >= 24); j0 = lane - 24 * r0
            put(FB_QB + r0 * 96 + 4 * j0, base.ravel(), (t0 + r0 * w) * 3 + 4 * j0)
        else:
            r0 = int(lane >= 56); j0 = lane - 48 - 8 * r0
            put(FB_QC + r0 * 32 + 4 * j0, cvis.ravel(), (t0 + r0 * w) * 1 + 4 * j0)
        # slot 1
        if lane < 16:
            r1 = int(lane >= 8); j1 = lane - 8 * r1
            put(FB_QL + r1 * 32 + 4 * j1, lvis.ravel(), (t0 + r1 * w) + 4 * j1)
        else:
            r1 = int(lane >= 40); j1 = lane - 16 - 24 * r1
            put(FB_QD + r1 * 96 + 4 * j1, dpred.ravel(), (t0 + r1 * w) * 3 + 4 * j1)
        # slot 2: dy1q
        tq = f * hw2 + y * w2 + 16 * xq
        put(FB_DYQ + 4 * lane, dy1q.ravel(), tq * 16 + 4 * lane)
        for io in range(K):
            fo = f * K + io
            o0 = fo * hw + 2 * y * w + 32 * xq
            O = FB_OBS + io * FB_O_SIZE
            a_rgb = lane < 48
            ra = int(lane >= 24) if a_rgb else 0
            ja = lane - 24 * ra if a_rgb else lane - 48
            put(O + (FB_O_RGB if a_rgb else FB_O_BASE) + ra * 96 + 4 * ja, (nn_rgb if a_rgb else nn_base).ravel(), (o0 + ra * w) * 3 + 4 * ja)
            if lane < 32:
                rb = int(lane >= 8); jb = 16 + lane if lane < 8 else lane - 8
                put(O + FB_O_BASE + rb * 96 + 4 * jb, nn_base.ravel(), (o0 + rb * w) * 3 + 4 * jb)
            put(O + FB_O_DY + 4 * lane, dy1o.ravel(), (fo * hw2 + y * w2 + 16 * xq) * 16 + 4 * lane)
    assert (written == 1).all(), "every float of the wave's LDS map is written exactly once"
    return L, (f, y, xq)


@pytest.mark.parametrize('n,h,w,K', [(1, 2, 32, 1), (2, 4, 64, 2), (1, 8, 96, 4), (3, 6, 32, 3)])
def test_staged_operands_are_the_gathered_ones(n, h, w, K):
    rng = np.random.default_rng(n + h + w + K)
    h2, w2 = h // 2, w // 2
    base, dpred = rng.random((n, h, w, 3)), rng.random((n, h, w, 3))
    cvis, lvis = rng.random((n, h, w)), rng.random((n, h, w))
    nn_rgb, nn_base = rng.random((n, K, h, w, 3)), rng.random((n, K, h, w, 3))
    dy1q, dy1o = rng.random((n, h2, w2, 16)), rng.random((n, K, h2, w2, 16))
    arrs = (base, cvis, lvis, dpred, nn_rgb, nn_base, dy1q, dy1o)
    runs = n * h2 * (w // 32)
    for run in range(runs):
        L, (f, y, xq) = stage_run(arrs, K, n, h, w, run)
        for g in range(4):
            for kk in range(4):
                xl = 4 * g + kk
                xh = 16 * xq + xl                                    # half-resolution column of this lane's texel
                for i in range(16):
                    assert L[FB_DYQ + xl * 16 + i] == dy1q[f, y, xh, i]
                    c, b = i & 7, i >> 3
                    for a in range(2):                               # query A operand: raw channel c at tap (a, b)
                        fy, fx, tx = 2 * y + a, 2 * xh + b, 2 * xl + b
                        if c < 3:
                            assert L[FB_QB + a * 96 + tx * 3 + c] == base[f, fy, fx, c]
                        elif c == 3:
                            assert L[FB_QC + a * 32 + tx] == cvis[f, fy, fx]
                        elif c == 4:
                            assert L[FB_QL + a * 32 + tx] == lvis[f, fy, fx]
                        else:
                            for io in range(K):
                                O = FB_OBS + io * FB_O_SIZE
                                assert L[O + FB_O_RGB + a * 96 + tx * 3 + c - 5] == nn_rgb[f, io, fy, fx, c - 5]
                                assert L[O + FB_O_BASE + a * 96 + tx * 3 + c - 5] == nn_base[f, io, fy, fx, c - 5]
                    if i < 12:                                       # R columns / observation A rows: (tap, channel)
                        otap, oc = divmod(i, 3)
                        fy, fx = 2 * y + (otap >> 1), 2 * xh + (otap & 1)
                        oo = (otap >> 1) * 96 + (2 * xl + (otap & 1)) * 3 + oc
                        assert L[FB_QD + oo] == dpred[f, fy, fx, oc]
                        for io in range(K):
                            O = FB_OBS + io * FB_O_SIZE
                            assert L[O + FB_O_RGB + oo] == nn_rgb[f, io, fy, fx, oc]
                            assert L[O + FB_O_BASE + oo] == nn_base[f, io, fy, fx, oc]
                    for io in range(K):
                        assert L[FB_OBS + io * FB_O_SIZE + FB_O_DY + xl * 16 + i] == dy1o[f, io, y, xh, i]
