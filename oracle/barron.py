"""Barron adaptive robust image loss as NLT uses it (fixed alpha=1, scale=0.01) -- CPU oracle.

ORACLE = test infrastructure (see oracle/__init__.py).  Restates
third_party/robust_loss/{wavelet.py:33-205,286-334,384-441, util.py:96-115,
general.py:89-123, distribution.py:88-114,149-222, cubic_spline.py:21-97,
adaptive.py:453-538} and nlt/losses.py:90-118.  PINNED against the reference's
golden data: tests/golden/wavelet_golden.npz (wavelet_test.py:146-171),
tests/golden/partition_spline.npz (distribution_test.py:86-106) and the closed
forms of general_test.py:245-257.
"""
import numpy as np
import torch

# wavelet.py:53-67  (CDF 9/7 half filters)
_CDF97_LO = np.array([+0.852698679009, +0.377402855613, -0.110624404418, -0.023849465020, +0.037828455507])
_CDF97_HI = np.array([+0.788485616406, -0.418092273222, -0.040689417609, +0.064538882629])
_mirror = lambda f: np.concatenate([f[-1:0:-1], f])          # wavelet.py:81
ANALYSIS_LO = _mirror(_CDF97_LO)                              # 9 taps
ANALYSIS_HI = _mirror(_CDF97_HI)                              # 7 taps

# robust_loss/util.py:96-97 and tf.image.rgb_to_yuv's kernel (TF source; rows R,G,B -> Y,U,V)
VOLUME_PRESERVING_YUV_SCALE = 1.580227820074
RGB_TO_YUV = np.array([[0.299, -0.14714119, 0.61497538],
                       [0.587, -0.28886916, -0.51496512],
                       [0.114, 0.43601035, -0.10001026]])

BARRON_ALPHA = 1.0      # nlt/losses.py:92
BARRON_SCALE = 0.01     # nlt/losses.py:94
BARRON_LEVELS = 5       # nlt/losses.py:103


def reflect_index(j, n):
    """wavelet.py:138-145 pad_reflecting's index map (unbounded reflections)."""
    period = max(1, 2 * (n - 1))
    jm = np.mod(j, period)
    return np.minimum(2 * (n - 1) - jm, jm)


def pad_reflecting(x, below, above, axis):
    n = x.shape[axis]
    j = reflect_index(np.arange(-below, n + above), n)
    return np.take(x, j, axis=axis)


def downsample_indices(n, flen, shift):
    """Index table for wavelet.py:164-205 `_downsample`: y[i]=sum_t f[t]*x[idx[i,t]]."""
    p = (flen - 1) // 2
    n_out = (n - 1 - shift) // 2 + 1
    i = np.arange(n_out)[:, None]
    t = np.arange(flen)[None, :]
    return reflect_index(2 * i + shift + t - p, n)


def _downsample(x, f, direction, shift):
    """x: torch (C, A, B); filter along axis direction+1 with reflection, stride 2."""
    axis = direction + 1
    idx = downsample_indices(x.shape[axis], len(f), shift)
    y = 0
    for t in range(len(f)):
        y = y + float(f[t]) * x.index_select(axis, torch.from_numpy(idx[:, t].astype(np.int64)))
    return y


def construct(im, num_levels):
    """wavelet.py:286-334 for 'CDF9/7'.  im: torch (C, A, B)."""
    pyr = []
    for _ in range(num_levels):
        hi = _downsample(im, ANALYSIS_HI, 0, 1)
        lo = _downsample(im, ANALYSIS_LO, 0, 0)
        pyr.append((_downsample(hi, ANALYSIS_HI, 1, 1),
                    _downsample(lo, ANALYSIS_HI, 1, 1),
                    _downsample(hi, ANALYSIS_LO, 1, 0)))
        im = _downsample(lo, ANALYSIS_LO, 1, 0)
    pyr.append(im)
    return tuple(pyr)


def flatten(pyr):
    """wavelet.py:408-441."""
    flat = pyr[-1]
    for d in range(len(pyr) - 2, -1, -1):
        flat = torch.cat([torch.cat([flat, pyr[d][1]], 2),
                          torch.cat([pyr[d][2], pyr[d][0]], 2)], 1)
    return flat


def rgb_to_syuv(rgb):
    """robust_loss/util.py:100-115."""
    m = torch.tensor(RGB_TO_YUV * VOLUME_PRESERVING_YUV_SCALE, dtype=rgb.dtype)
    return rgb @ m


# ---- general.py:29-125 (exact branch), NumPy, all alphas: used only to pin closed forms
def lossfun(x, alpha, scale):
    x = np.asarray(x, np.float64)
    alpha = np.broadcast_to(np.asarray(alpha, np.float64), x.shape)
    scale = np.broadcast_to(np.asarray(scale, np.float64), x.shape)
    sq = np.square(x / scale)
    eps = np.float64(np.finfo(np.float32).eps)
    loss_two = 0.5 * sq
    loss_zero = np.log1p(np.minimum(0.5 * sq, 3e37))
    loss_neginf = -np.expm1(-0.5 * sq)
    loss_posinf = np.expm1(np.minimum(0.5 * sq, 87.5))
    beta_safe = np.maximum(eps, np.abs(alpha - 2.))
    alpha_safe = np.where(alpha >= 0, 1., -1.) * np.maximum(eps, np.abs(alpha))
    with np.errstate(all='ignore'):
        otherwise = (beta_safe / alpha_safe) * (np.power(sq / beta_safe + 1., 0.5 * alpha) - 1.)
    return np.where(alpha == -np.inf, loss_neginf,
                    np.where(alpha == 0, loss_zero,
                             np.where(alpha == 2, loss_two,
                                      np.where(alpha == np.inf, loss_posinf, otherwise))))


# ---- distribution.py:88-114 / cubic_spline.py:21-97 / distribution.py:149-179
def partition_spline_curve(alpha):
    alpha = np.asarray(alpha, np.float64)
    with np.errstate(all='ignore'):
        return np.where(alpha < 4,
                        (2.25 * alpha - 4.5) / (np.abs(alpha - 2) + 0.25) + alpha + 2,
                        5. / 18. * np.log(np.clip(4 * alpha - 15, 1e-300, 3e37)) + 8)   # util.log_safe clamp


def interpolate1d(x, values, tangents):
    x = np.asarray(x, np.float64)
    x_lo = np.floor(np.clip(x, 0., len(values) - 2)).astype(np.int64)
    x_hi = x_lo + 1
    t = x - x_lo
    t_sq = t * t; t_cu = t * t_sq
    h01 = -2. * t_cu + 3. * t_sq
    h00 = 1. - h01
    h11 = t_cu - t_sq
    h10 = h11 - t_sq + t
    before = tangents[0] * t + values[0]
    after = tangents[-1] * (t - 1.) + values[-1]
    mid = values[x_lo] * h00 + values[x_hi] * h01 + tangents[x_lo] * h10 + tangents[x_hi] * h11
    return np.where(t < 0., before, np.where(t > 1., after, mid))


def log_base_partition_function(alpha, spline):
    x = partition_spline_curve(alpha)
    return interpolate1d(x * float(spline['x_scale']), spline['values'], spline['tangents'])


# log Z(alpha=1) from the reference's spline (tests/golden/partition_spline.npz); analytically
# log(2 e K_1(1)).  tests/test_oracle_barron.py re-derives it from the npz and from scipy.special.
LOG_Z_ALPHA1 = 1.1854952325


def charbonnier_nll(w, scale=BARRON_SCALE, log_z=LOG_Z_ALPHA1):
    """distribution.py:181-222 with alpha=1: rho + log(scale) + logZ(1);
    rho = sqrt((w/c)^2+1)-1 (general.py:104-112 with beta_safe=alpha_safe=1)."""
    return torch.sqrt((w / scale) ** 2 + 1.) - 1. + float(np.log(scale) + log_z)


def barron_loss(gt, pred, keep_batch=False, weights=None):
    """nlt/losses.py:107-118 -> adaptive.py:453-538 (color_space='YUV',
    representation='CDF9/7', 5 levels, wavelet_scale_base=1 => rescale is a no-op).
    weights: gt and pred alpha-blended against zeros first (losses.py:107-110, nlt/util/img.py:74-89)."""
    if weights is not None:
        alpha = torch.as_tensor(weights, dtype=gt.dtype)
        gt = gt * alpha + torch.zeros_like(gt) * (1 - alpha)
        pred = pred * alpha + torch.zeros_like(pred) * (1 - alpha)
    x = gt - pred                                       # losses.py:111
    n, h, w, c = x.shape
    x = rgb_to_syuv(x)                                  # adaptive.py:478-479
    stack = x.permute(0, 3, 1, 2).reshape(n * c, h, w)  # adaptive.py:486-487
    flat = flatten(construct(stack, BARRON_LEVELS))     # adaptive.py:492-496
    nll = charbonnier_nll(flat).reshape(n, c, h, w)     # adaptive.py:504-514 (elementwise)
    return nll.mean(dim=(1, 2, 3)) if keep_batch else nll.mean()
