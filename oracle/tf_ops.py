"""TensorFlow / Keras / TF-Addons primitives restated on torch-CPU (+ naive NumPy twins).

ORACLE = test infrastructure (see oracle/__init__.py).  Every function cites the
reference call site it stands in for.  All tensors are NHWC like the reference.

The torch versions are differentiable (used for train_step parity through
torch-CPU autograd); the `*_naive` NumPy versions are literal loops written from
the documented TF semantics and exist only so that the padding / layout claims of
the torch versions are themselves tested (tests/test_oracle_ops.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

LRELU_ALPHA = 0.3  # tf.keras.layers.LeakyReLU(alpha=0.3): nlt/networks/elements.py:72-73


# ----------------------------------------------------------------------------
# Keras Conv2D(padding='same')            nlt/networks/elements.py:26-31
# ----------------------------------------------------------------------------
def _same_pad(in_size, k, s):
    """TF 'SAME': out=ceil(in/s); pad_total=max((out-1)*s+k-in,0); before=total//2."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


def conv2d_same(x, w_hwio, b, stride):
    """x [N,H,W,Cin]; w_hwio [kh,kw,Cin,Cout] (Keras layout); b [Cout] -> [N,H/s,W/s,Cout].

    Cross-correlation (TF conv2d does not flip).  k=2,s=1 pads bottom/right by 1;
    k=2,s=2 on even sizes pads nothing (SURVEY a-C2/a-C3)."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    pt, pb = _same_pad(x.shape[1], kh, stride)
    pl, pr = _same_pad(x.shape[2], kw, stride)
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn, w_hwio.permute(3, 2, 0, 1), b, stride=stride)
    return y.permute(0, 2, 3, 1)


def conv2d_same_naive(x, w, b, stride):
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64)
    n, h, wd, _ = x.shape
    kh, kw, _, co = w.shape
    oh, ow = -(-h // stride), -(-wd // stride)
    pt, _ = _same_pad(h, kh, stride)
    pl, _ = _same_pad(wd, kw, stride)
    y = np.zeros((n, oh, ow, co))
    for i in range(oh):
        for j in range(ow):
            acc = np.zeros((n, co))
            for a in range(kh):
                for bb in range(kw):
                    yi, xi = i * stride + a - pt, j * stride + bb - pl
                    if 0 <= yi < h and 0 <= xi < wd:
                        acc += x[:, yi, xi, :] @ w[a, bb]
            y[:, i, j, :] = acc + np.asarray(b, np.float64)
    return y


# ----------------------------------------------------------------------------
# Keras Conv2DTranspose(padding='same')   nlt/networks/elements.py:34-39
# ----------------------------------------------------------------------------
def conv2d_transpose_same(x, w_hwoi, b, stride):
    """x [N,h,w,Cin]; w_hwoi [kh,kw,Cout,Cin] (Keras layout) -> [N,h*s,w*s,Cout].

    = gradient of the SAME-padded forward conv.  k=2,s=2: each input texel owns a
    2x2 output block.  k=2,s=1: y[i,j]=sum_{a,b} x[i-a,j-b] W[a,b], x[-1]=0, i.e.
    the full transposed conv cropped to its first h x w (SURVEY a-D2/a-D3)."""
    n, h, wd, _ = x.shape
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w_hwoi.permute(3, 2, 0, 1), b, stride=stride)
    y = y[:, :, : h * stride, : wd * stride]
    return y.permute(0, 2, 3, 1)


def conv2d_transpose_same_naive(x, w, b, stride):
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64)
    n, h, wd, _ = x.shape
    kh, kw, co, _ = w.shape
    full = np.zeros((n, (h - 1) * stride + kh, (wd - 1) * stride + kw, co))
    for i in range(h):
        for j in range(wd):
            for a in range(kh):
                for bb in range(kw):
                    full[:, i * stride + a, j * stride + bb, :] += x[:, i, j, :] @ w[a, bb].T
    # forward SAME conv pads (before=total//2): k2s1 -> (0,1), k2s2 -> (0,0); its
    # transpose therefore crops `before` rows from the top, i.e. none.
    return full[:, : h * stride, : wd * stride, :] + np.asarray(b, np.float64)


def leaky_relu(x, alpha=LRELU_ALPHA):
    return F.leaky_relu(x, alpha)


# ----------------------------------------------------------------------------
# tfa.image.resampler (TF-Addons 0.10.0)  nlt/models/nlt.py:112-114
# ----------------------------------------------------------------------------
def resampler_indices(warp_xy, h, w):
    """Integer corner indices + validity, the bit-exact part of the gather.

    warp_xy: float32 array [...,2] in PIXEL units (x -> width, y -> height).
    Returns int32 fx, fy (floor; cx=fx+1, cy=fy+1) and the `inside` predicate
    x>-1 & y>-1 & x<w & y<h, all computed in float32 like the TFA kernel."""
    warp_xy = np.asarray(warp_xy, np.float32)
    x, y = warp_xy[..., 0], warp_xy[..., 1]
    inside = (x > np.float32(-1)) & (y > np.float32(-1)) & (x < np.float32(w)) & (y < np.float32(h))
    fx = np.floor(x).astype(np.int32)
    fy = np.floor(y).astype(np.int32)
    return fx, fy, inside


def resampler_naive(data, warp_xy):
    """Literal restatement of the TFA resampler CPU kernel, float32 arithmetic.

    out = dx*dy*D(fx,fy) + (1-dx)(1-dy)*D(cx,cy) + dx(1-dy)*D(fx,cy) + (1-dx)dy*D(cx,fy)
    with dx=cx-x, dy=cy-y and D(.)=0 for integer indices outside the image; the
    whole sample is 0 unless x>-1, y>-1, x<W, y<H."""
    data = np.asarray(data, np.float32)
    warp_xy = np.asarray(warp_xy, np.float32)
    n, h, w, c = data.shape
    out = np.zeros(warp_xy.shape[:-1] + (c,), np.float32)
    fx, fy, inside = resampler_indices(warp_xy, h, w)
    one = np.float32(1)

    def D(b, xi, yi):
        if xi < 0 or yi < 0 or xi > w - 1 or yi > h - 1:
            return np.zeros(c, np.float32)
        return data[b, yi, xi]

    it = np.ndindex(*warp_xy.shape[:-1])
    for idx in it:
        if not inside[idx]:
            continue
        b = idx[0]
        x, y = warp_xy[idx]
        fx_, fy_ = int(fx[idx]), int(fy[idx])
        cx_, cy_ = fx_ + 1, fy_ + 1
        dx = np.float32(cx_) - x
        dy = np.float32(cy_) - y
        v = (dx * dy) * D(b, fx_, fy_)
        v = v + ((one - dx) * (one - dy)) * D(b, cx_, cy_)
        v = v + (dx * (one - dy)) * D(b, fx_, cy_)
        v = v + ((one - dx) * dy) * D(b, cx_, fy_)
        out[idx] = v
    return out


def resampler(data, warp_xy):
    """Differentiable torch twin of resampler_naive (grad w.r.t. data = 4-corner
    scatter-add with the same weights, as in the TFA gradient kernel)."""
    n, h, w, c = data.shape
    x, y = warp_xy[..., 0], warp_xy[..., 1]
    inside = (x > -1) & (y > -1) & (x < w) & (y < h)
    fx = torch.floor(x); fy = torch.floor(y)
    cx = fx + 1; cy = fy + 1
    dx = cx - x; dy = cy - y
    flat = data.reshape(n, h * w, c)

    def D(xi, yi):
        ok = (xi >= 0) & (yi >= 0) & (xi <= w - 1) & (yi <= h - 1) & inside
        lin = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).long().reshape(n, -1)
        g = torch.gather(flat, 1, lin.unsqueeze(-1).expand(-1, -1, c))
        return g.reshape(*x.shape, c) * ok.unsqueeze(-1).to(data.dtype)

    out = (dx * dy).unsqueeze(-1) * D(fx, fy)
    out = out + ((1 - dx) * (1 - dy)).unsqueeze(-1) * D(cx, cy)
    out = out + (dx * (1 - dy)).unsqueeze(-1) * D(fx, cy)
    out = out + ((1 - dx) * dy).unsqueeze(-1) * D(cx, fy)
    return out


# ----------------------------------------------------------------------------
# tf.image.resize (TF2 default)           nlt/util/img.py:92-120
# ----------------------------------------------------------------------------
def resize_bilinear(x, new_h, new_w):
    """Bilinear, half_pixel_centers=True, antialias=False; identity when equal."""
    if x.shape[1] == new_h and x.shape[2] == new_w:
        return x
    y = F.interpolate(x.permute(0, 3, 1, 2), size=(new_h, new_w), mode='bilinear',
                      align_corners=False, antialias=False)
    return y.permute(0, 2, 3, 1)


def resize_bilinear_naive(x, new_h, new_w):
    """TF2 ResizeBilinear with half-pixel centres (compute_interpolation_weights):
    src=(dst+0.5)*scale-0.5; lo=max(floor(src),0); hi=min(ceil(src),S-1);
    lerp=src-floor(src)."""
    x = np.asarray(x, np.float32)
    n, h, w, c = x.shape

    def weights(out_size, in_size):
        scale = np.float32(in_size) / np.float32(out_size)
        lo = np.zeros(out_size, np.int64); hi = np.zeros(out_size, np.int64)
        lerp = np.zeros(out_size, np.float32)
        for i in range(out_size):
            src = (np.float32(i) + np.float32(0.5)) * scale - np.float32(0.5)
            fl = np.floor(src)
            lo[i] = max(int(fl), 0)
            hi[i] = min(int(np.ceil(src)), in_size - 1)
            lerp[i] = src - fl
        return lo, hi, lerp

    ylo, yhi, yl = weights(new_h, h)
    xlo, xhi, xl = weights(new_w, w)
    out = np.zeros((n, new_h, new_w, c), np.float32)
    for i in range(new_h):
        for j in range(new_w):
            tl = x[:, ylo[i], xlo[j]]; tr = x[:, ylo[i], xhi[j]]
            bl = x[:, yhi[i], xlo[j]]; br = x[:, yhi[i], xhi[j]]
            top = tl + (tr - tl) * xl[j]
            bot = bl + (br - bl) * xl[j]
            out[:, i, j] = top + (bot - top) * yl[i]
    return out


# ----------------------------------------------------------------------------
# small elementwise helpers                nlt/util/img.py:74-89,179-185
# ----------------------------------------------------------------------------
def set_left_top_corner(x, val):
    mask = torch.ones_like(x)
    mask[:, 0, 0, :] = val
    return mask * x


def alpha_blend(t1, alpha):
    return t1 * alpha  # second tensor None -> zeros (util/img.py:87-89)


# ----------------------------------------------------------------------------
# Keras initialisers (Conv2D defaults: glorot_uniform kernel, zero bias)
# ----------------------------------------------------------------------------
def glorot_uniform(rng, shape_hw_a_b):
    """Keras glorot_uniform for a conv kernel (kh,kw,A,B): fan_in=kh*kw*A,
    fan_out=kh*kw*B, limit=sqrt(6/(fan_in+fan_out)).  For Conv2DTranspose the
    kernel is (kh,kw,Cout,Cin) and Keras applies the same rule to that shape."""
    kh, kw, a, b = shape_hw_a_b
    limit = np.sqrt(6.0 / (kh * kw * a + kh * kw * b))
    return rng.uniform(-limit, limit, size=shape_hw_a_b).astype(np.float32)


# ----------------------------------------------------------------------------
# bf16 1x1 channel mix (BASELINE config 5): Conv2D(kernel_size=1) of
# nlt/networks/elements.py:26-31 on bf16-rounded operands, fp32 accumulation
# ----------------------------------------------------------------------------
def conv1x1_bf16(x_bf16, w_hwio, b, act=True, alpha=LRELU_ALPHA):
    """x [...,Cin] bfloat16; w_hwio (1,1,Cin,Cout) fp32 (rounded to bf16 here, round-to-nearest-even, as the
    kernel's packer does); b fp32.  Products exact in fp32, sum in fp32 (float64 here: the summation order is the
    kernel's business), bias, LeakyReLU, one final rounding to bf16."""
    w = w_hwio[0, 0].to(torch.bfloat16).to(torch.float64)
    y = x_bf16.to(torch.float64) @ w + b.to(torch.float64)
    if act:
        y = torch.where(y > 0, y, alpha * y)
    return y.to(torch.float32).to(torch.bfloat16)
