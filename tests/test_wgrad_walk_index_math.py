"""CPU: NumPy emulation of the ADDRESS WALK that is new in csrc/wgrad_tile.hip (wgrad_walk_kernel) and in the WALK form of
csrc/wgrad_narrow.hip: a wave's run of rows, 4 per step, with wave-uniform (x0, y), lane pointers advanced by two constants
(plain step / step that wraps to the next image row) and padding validity from lane-constant bounds.  Every step must read exactly
the texels nlt_common.h's conv_tap_texel() derives from the row index -- for all five conv families, several frames, narrow grids,
runs that start in the middle of an image row and runs clipped by M -- and must never form an address outside the tensors."""
import numpy as np
import pytest

CONV1X1, CONV_K2S2, CONV_K2S1, DECONV_K2S2, DECONV_K2S1 = range(5)


def tap_texel(mode, h, w, f, y, x, t):
    """conv_tap_texel (nlt_common.h:45-56): input texel of tap t = (a, b) of GEMM row (f, y, x), or -1 in the zero padding."""
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


def walk(mode, n, h, w, m0, chunk, tap, ab):
    """The kernel's bookkeeping for ONE lane group kk = 0..3 of one wave: yields per step, per kk:
    (A texel or -1, B texel).  h, w = input dims of the layer (as nlt_fill_conv_params takes them)."""
    gh, gw = (h // 2, w // 2) if mode == CONV_K2S2 else (h, w)
    oh, ow = (2 * h, 2 * w) if mode == DECONV_K2S2 else (gh, gw)
    M = n * gh * gw
    assert gw % 4 == 0 and m0 % 4 == 0 and chunk % 4 == 0
    m1 = min(m0 + chunk, M)
    nsteps = max(0, (m1 - m0) // 4)
    ta, tb = tap >> 1, tap & 1
    out = []
    x0 = m0 % gw
    R0 = m0 // gw
    y = R0 % gh
    a_tex, b_tex = [], []
    for kk in range(4):
        if mode == CONV_K2S2:
            a_tex.append((2 * R0 + ta) * w + 2 * (x0 + kk) + tb)
        elif mode == CONV_K2S1:
            a_tex.append(m0 + kk + ta * w + tb)
        elif mode == DECONV_K2S1:
            a_tex.append(m0 + kk - ta * w - tb)
        else:
            a_tex.append(m0 + kk)
        if mode == DECONV_K2S2:
            b_tex.append((2 * R0 + (ab >> 1)) * ow + 2 * (x0 + kk) + (ab & 1))
        else:
            b_tex.append(m0 + kk)
    a_inc = 8 if mode == CONV_K2S2 else 4
    a_inc_w = 8 + w if mode == CONV_K2S2 else a_inc
    b_inc = 8 if mode == DECONV_K2S2 else 4
    b_inc_w = 8 + ow if mode == DECONV_K2S2 else b_inc
    issued = 0
    for _ in range(nsteps):
        row = []
        for kk in range(4):
            ok = True
            if mode == CONV_K2S1:
                ok = x0 < gw - kk - tb and y < gh - ta
            if mode == DECONV_K2S1:
                ok = x0 >= tb - kk and y >= ta
            row.append((a_tex[kk] if ok else -1, b_tex[kk]))
        out.append(row)
        issued += 1
        if issued < nsteps:
            xn = x0 + 4
            wrap = xn >= gw
            for kk in range(4):
                a_tex[kk] += a_inc_w if wrap else a_inc
                b_tex[kk] += b_inc_w if wrap else b_inc
            if wrap:
                x0, y = 0, (0 if y + 1 >= gh else y + 1)
            else:
                x0 = xn
    return out, m0, nsteps, (gh, gw, oh, ow)


@pytest.mark.parametrize('mode', [CONV1X1, CONV_K2S2, CONV_K2S1, DECONV_K2S2, DECONV_K2S1])
@pytest.mark.parametrize('n,h,w', [(1, 8, 8), (3, 8, 16), (2, 4, 4), (2, 16, 8), (1, 2, 32)])
def test_walk_visits_the_texels_the_row_index_derives(mode, n, h, w):
    gh, gw = (h // 2, w // 2) if mode == CONV_K2S2 else (h, w)
    if gw % 4:
        pytest.skip("the walk form needs a row grid that is a multiple of 4 texels wide (the generic kernel takes the rest)")
    M = n * gh * gw
    taps = (0,) if mode in (CONV1X1, DECONV_K2S2) else (0, 1, 2, 3)
    abs_ = (0, 1, 2, 3) if mode == DECONV_K2S2 else (0,)
    in_texels = n * h * w
    for chunk in (4, 24, 96):
        for m0 in range(0, M + chunk, chunk):                        # includes a run that starts past M (empty) and clipped ones
            for tap in taps:
                for ab in abs_:
                    steps, _, nsteps, (gh_, gw_, oh, ow) = walk(mode, n, h, w, m0, chunk, tap, ab)
                    assert nsteps == max(0, (min(m0 + chunk, M) - m0) // 4)
                    for s, row in enumerate(steps):
                        for kk, (at, bt) in enumerate(row):
                            m = m0 + 4 * s + kk
                            f, r = divmod(m, gh * gw)
                            y, x = divmod(r, gw)
                            assert at == tap_texel(mode, h, w, f, y, x, tap), (mode, m, tap)
                            if at >= 0:
                                assert at < in_texels
                            want_b = (f * oh + 2 * y + (ab >> 1)) * ow + 2 * x + (ab & 1) if mode == DECONV_K2S2 else m
                            assert bt == want_b and 0 <= bt < n * oh * ow
