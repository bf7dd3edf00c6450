"""TEST-ONLY stand-in for the Python-level `nlt_amd._capi` adapters, executing each C-ABI op
with the oracle's torch-CPU primitives on CPU tensors.  It honours the pointer/stride calling
convention (channel slices of wider tensors), so the host orchestration (engine buffers,
virtual concat, slices, labels) can be verified without a GPU.  Never used by the product."""
import torch

from nlt_amd import _capi as C
from oracle import tf_ops as T

_MODES = {C.CONV1X1: (1, False), C.CONV_K2S2: (2, False), C.CONV_K2S1: (1, False),
          C.DECONV_K2S2: (2, True), C.DECONV_K2S1: (1, True)}


def _view(t, n, h, w, c, ld):
    return torch.as_strided(t, (n, h, w, c), (h * w * ld, w * ld, ld, 1))


def conv_forward(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, w_keras, w_packed, bias, cout, out, ldo,
                 act=True, alpha=0.3, algo=0, tile_hint=0, mask_src=None, ldm=0, accumulate=False):
    x = _view(src0, n, h, w, c0, ld0)
    if c1:
        x = torch.cat((x, _view(src1, n, h, w, c1, ld1)), -1)
    s, tr = _MODES[mode]
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    y = f(x, w_keras, bias, s)
    oh, ow = y.shape[1:3]
    o = _view(out, n, oh, ow, cout, ldo)
    if accumulate:
        y = y + o
    if mask_src is not None:
        y = y * torch.where(_view(mask_src, n, oh, ow, cout, ldm) > 0, 1.0, alpha)
    elif act:
        y = T.leaky_relu(y, alpha)
    o.copy_(y)


def pack_conv_weights(mode, w_keras, c0, c1, cout):
    return torch.zeros(1)


def stem_forward(base, cvis, lvis, nn_rgb, nn_base, obs_weights, n, k, h, w, c, wq, bq, wo, bo, fm0, obs0):
    x = torch.cat((base, cvis, lvis), -1)
    fm0[..., :c] = x @ wq[0, 0] + bq
    o = (nn_rgb - nn_base) @ wo[0, 0] + bo
    obs0.copy_(o)
    if obs_weights is not None:
        o = o * obs_weights[:, :, None, None, None]
    fm0[..., c:] = o.mean(1)


def obs_mean_forward(obs, obs_weights, n, k, hw, c, out, ldo):
    o = obs.reshape(n, k, hw, c)
    if obs_weights is not None:
        o = o * obs_weights[:, :, None, None]
    torch.as_strided(out, (n, hw, c), (hw * ldo, ldo, 1)).copy_(o.mean(1))


def head_forward(dec, ldd, cd, skip, lds, cs, w_keras, bias, base, n, h, w, pred):
    x = _view(dec, n, h, w, cd, ldd)
    if cs:
        x = torch.cat((x, _view(skip, n, h, w, cs, lds)), -1)
    y = x @ w_keras[0, 0] + bias
    if base is not None:
        y = y + base
    y[:, 0, 0, :] = 0
    pred.copy_(y)


def warp_forward(pred, base, warp, n, uvh, uvw, hc, wc, pred_cam, base_cam, fg_cam, idx_out=None):
    wpx = warp * torch.tensor([uvw, uvh], dtype=torch.float32)
    if pred_cam is not None:
        pred_cam.copy_(T.resampler(pred, wpx))
    if base_cam is not None:
        base_cam.copy_(T.resampler(T.set_left_top_corner(base, 0), wpx))
    if fg_cam is not None:
        fg_cam.copy_(T.resampler(T.set_left_top_corner(torch.ones_like(pred), 0), wpx))
    if idx_out is not None:
        fx, fy, inside = T.resampler_indices(wpx.numpy(), uvh, uvw)
        idx_out[..., 0] = torch.from_numpy(fx); idx_out[..., 1] = torch.from_numpy(fy)
        idx_out[..., 2] = torch.from_numpy(inside.astype('int32')); idx_out[..., 3] = 0


def resize_bilinear_forward(x, oh, ow):
    return T.resize_bilinear(x, oh, ow).contiguous()


def mul_forward(a, b):
    return a * b


def install(monkeypatch):
    for name in ('conv_forward', 'pack_conv_weights', 'stem_forward', 'obs_mean_forward', 'head_forward',
                 'warp_forward', 'resize_bilinear_forward', 'mul_forward'):
        monkeypatch.setattr(C, name, globals()[name])
