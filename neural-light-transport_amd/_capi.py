"""ctypes binding of libnlt_hip.so (include/nlt_hip.h) + thin torch-tensor adapters.

PyTorch is plumbing here (device memory + streams): every function takes torch CUDA tensors,
checks dtype/contiguity, and hands raw device pointers and the current HIP stream to the
C ABI.  There is NO CPU or torch fallback: if the library is missing or a call returns a
non-zero status this raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libnlt_hip.so')

CONV1X1, CONV_K2S2, CONV_K2S1, DECONV_K2S2, DECONV_K2S1 = range(5)
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA = range(3)

_c_int, _c_long, _c_float, _vp = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/nlt_hip.h declares
SIGNATURES = {
    'nlt_version': (ctypes.c_char_p, []),
    'nlt_status_string': (ctypes.c_char_p, [_c_int]),
    'nlt_packed_weight_floats': (_c_long, [_c_int] * 4),
    'nlt_pack_conv_weights': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_conv_forward': (_c_int, [_c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _c_int,
                                  _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _vp, _c_int,
                                  _c_int, _c_float, _vp, _c_int, _c_int, _vp]),
    'nlt_stem_forward': (_c_int, [_vp] * 6 + [_c_int] * 5 + [_vp] * 4 + [_vp, _vp, _vp]),
    'nlt_obs_mean_forward': (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp]),
    'nlt_head_forward': (_c_int, [_vp, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _vp,
                                  _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_warp_forward': (_c_int, [_vp, _vp, _vp] + [_c_int] * 5 + [_vp] * 5),
    'nlt_resize_bilinear_forward': (_c_int, [_vp] + [_c_int] * 6 + [_vp, _vp]),
    'nlt_mul_forward': (_c_int, [_vp, _vp, _c_long, _vp, _vp]),
    'nlt_conv_backward_weights': (_c_int, [_c_int, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _c_int,
                                           _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _vp]),
    'nlt_lrelu_backward': (_c_int, [_vp, _c_int, _vp, _c_int, _c_int, _c_long, _c_float, _vp, _c_int, _vp]),
    'nlt_obs_mean_backward': (_c_int, [_vp, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _vp, _vp]),
    'nlt_stem_backward': (_c_int, [_vp] * 6 + [_c_int] * 5 + [_vp] * 6 + [_vp]),
    'nlt_head_backward': (_c_int, [_vp, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int,
                                   _vp, _c_int, _vp, _c_int, _vp, _vp, _vp]),
    'nlt_warp_backward': (_c_int, [_vp, _vp] + [_c_int] * 5 + [_vp, _vp]),
    'nlt_resize_bilinear_backward': (_c_int, [_vp] + [_c_int] * 6 + [_vp, _vp]),
    'nlt_l2_loss_forward': (_c_int, [_vp, _vp, _c_int, _c_long, _vp, _vp]),
    'nlt_l2_loss_backward': (_c_int, [_vp, _vp, _vp, _c_int, _c_long, _vp, _vp]),
    'nlt_barron_workspace_floats': (_c_long, [_c_int] * 3),
    'nlt_barron_loss': (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp]),
    'nlt_scale_rows': (_c_int, [_vp, _vp, _c_int, _c_long, _vp, _vp]),
    'nlt_adam_amsgrad_step': (_c_int, [_vp] * 5 + [_c_long] + [_c_float] * 4 + [_vp]),
}

_lib = None


def lib():
    """Loads libnlt_hip.so once; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libnlt_hip.so not found at %s -- build it with `python __graft_entry__.py` or "
                "`make -C neural-light-transport_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class NLTError(RuntimeError):
    pass


def _check(status, what):
    if status != 0:
        raise NLTError("%s failed: %s (%d)" % (what, lib().nlt_status_string(status).decode(), status))


def _ptr(t):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype in (torch.float32, torch.int32)):
        raise NLTError("expected a float32/int32 CUDA tensor, got %s on %s" % (t.dtype, t.device))
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dense(t, what):
    if not t.is_contiguous():
        raise NLTError("%s must be contiguous" % what)
    return t


def packed_weight_floats(mode, c0, c1, cout):
    return lib().nlt_packed_weight_floats(mode, c0, c1, cout)


def pack_conv_weights(mode, w_keras, c0, c1, cout):
    n = packed_weight_floats(mode, c0, c1, cout)
    out = torch.empty(n, device=w_keras.device, dtype=torch.float32)
    _check(lib().nlt_pack_conv_weights(mode, _ptr(_dense(w_keras, 'w_keras')), c0, c1, cout, _ptr(out), _stream()),
           'nlt_pack_conv_weights')
    return out


def conv_forward(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, w_keras, w_packed, bias, cout, out, ldo,
                 act=True, alpha=0.3, algo=ALGO_AUTO, tile_hint=0, mask_src=None, ldm=0, accumulate=False):
    """src*/out/mask_src may be views INTO wider tensors: pass the (already offset) tensor whose
    data_ptr() is the first element of the channel slice, and the per-texel stride ld*."""
    _check(lib().nlt_conv_forward(mode, algo, tile_hint, _ptr(src0), ld0, c0, _ptr(src1), ld1, c1, n, h, w,
                                  _ptr(w_keras), _ptr(w_packed), _ptr(bias), cout, _ptr(out), ldo,
                                  1 if act else 0, float(alpha), _ptr(mask_src), ldm, 1 if accumulate else 0,
                                  _stream()), 'nlt_conv_forward')


def stem_forward(base, cvis, lvis, nn_rgb, nn_base, obs_weights, n, k, h, w, c, wq, bq, wo, bo, fm0, obs0):
    for t, nm in ((base, 'base'), (cvis, 'cvis'), (lvis, 'lvis'), (nn_rgb, 'nn_rgb'), (nn_base, 'nn_base')):
        _dense(t, nm)
    _check(lib().nlt_stem_forward(_ptr(base), _ptr(cvis), _ptr(lvis), _ptr(nn_rgb), _ptr(nn_base),
                                  _ptr(obs_weights), n, k, h, w, c, _ptr(wq), _ptr(bq), _ptr(wo), _ptr(bo),
                                  _ptr(fm0), _ptr(obs0), _stream()), 'nlt_stem_forward')


def obs_mean_forward(obs, obs_weights, n, k, hw, c, out, ldo):
    _check(lib().nlt_obs_mean_forward(_ptr(_dense(obs, 'obs')), _ptr(obs_weights), n, k, hw, c, _ptr(out), ldo,
                                      _stream()), 'nlt_obs_mean_forward')


def head_forward(dec, ldd, cd, skip, lds, cs, w_keras, bias, base, n, h, w, pred):
    _check(lib().nlt_head_forward(_ptr(dec), ldd, cd, _ptr(skip), lds, cs, _ptr(w_keras), _ptr(bias), _ptr(base),
                                  n, h, w, _ptr(pred), _stream()), 'nlt_head_forward')


def warp_forward(pred, base, warp, n, uvh, uvw, hc, wc, pred_cam, base_cam, fg_cam, idx_out=None):
    _check(lib().nlt_warp_forward(_ptr(pred), _ptr(base), _ptr(_dense(warp, 'warp')), n, uvh, uvw, hc, wc,
                                  _ptr(pred_cam), _ptr(base_cam), _ptr(fg_cam), _ptr(idx_out), _stream()),
           'nlt_warp_forward')


def resize_bilinear_forward(x, oh, ow):
    n, h, w, c = x.shape
    out = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
    _check(lib().nlt_resize_bilinear_forward(_ptr(_dense(x, 'x')), n, h, w, c, oh, ow, _ptr(out), _stream()),
           'nlt_resize_bilinear_forward')
    return out


def mul_forward(a, b):
    out = torch.empty_like(a)
    _check(lib().nlt_mul_forward(_ptr(_dense(a, 'a')), _ptr(_dense(b, 'b')), a.numel(), _ptr(out), _stream()),
           'nlt_mul_forward')
    return out


# ---------------------------------------------------------------- train step
def conv_backward_weights(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp, cout, dw, db, algo=ALGO_AUTO):
    _check(lib().nlt_conv_backward_weights(mode, algo, _ptr(src0), ld0, c0, _ptr(src1), ld1, c1, n, h, w,
                                           _ptr(dpre), ldp, cout, _ptr(dw), _ptr(db), _stream()),
           'nlt_conv_backward_weights')


def lrelu_backward(g, ldg, y, ldy, c, texels, alpha, out, ldo):
    _check(lib().nlt_lrelu_backward(_ptr(g), ldg, _ptr(y), ldy, c, texels, float(alpha), _ptr(out), ldo, _stream()),
           'nlt_lrelu_backward')


def obs_mean_backward(dmean, ldm, obs_y, obs_weights, dobs_partial, n, k, hw, c, alpha, dpre_obs):
    _check(lib().nlt_obs_mean_backward(_ptr(dmean), ldm, _ptr(obs_y), _ptr(obs_weights), _ptr(dobs_partial),
                                       n, k, hw, c, float(alpha), _ptr(dpre_obs), _stream()), 'nlt_obs_mean_backward')


def stem_backward(base, cvis, lvis, nn_rgb, nn_base, obs_weights, n, k, h, w, c, dfm0, dobs0, dwq, dbq, dwo, dbo):
    _check(lib().nlt_stem_backward(_ptr(base), _ptr(cvis), _ptr(lvis), _ptr(nn_rgb), _ptr(nn_base), _ptr(obs_weights),
                                   n, k, h, w, c, _ptr(dfm0), _ptr(dobs0), _ptr(dwq), _ptr(dbq), _ptr(dwo), _ptr(dbo),
                                   _stream()), 'nlt_stem_backward')


def head_backward(dec, ldd, cd, skip, lds, cs, w_keras, dpred, n, h, w, d_dec, ldgd, d_skip, ldgs, dw, db):
    _check(lib().nlt_head_backward(_ptr(dec), ldd, cd, _ptr(skip), lds, cs, _ptr(w_keras), _ptr(_dense(dpred, 'dpred')),
                                   n, h, w, _ptr(d_dec), ldgd, _ptr(d_skip), ldgs, _ptr(dw), _ptr(db), _stream()),
           'nlt_head_backward')


def warp_backward(dpred_cam, warp, n, uvh, uvw, hc, wc, dpred):
    _check(lib().nlt_warp_backward(_ptr(_dense(dpred_cam, 'dpred_cam')), _ptr(_dense(warp, 'warp')), n, uvh, uvw, hc, wc,
                                   _ptr(dpred), _stream()), 'nlt_warp_backward')


def resize_bilinear_backward(dout, h, w):
    n, oh, ow, c = dout.shape
    dx = torch.empty((n, h, w, c), device=dout.device, dtype=torch.float32)
    _check(lib().nlt_resize_bilinear_backward(_ptr(_dense(dout, 'dout')), n, h, w, c, oh, ow, _ptr(dx), _stream()),
           'nlt_resize_bilinear_backward')
    return dx


def l2_loss_forward(pred, gt):
    n = pred.shape[0]
    loss = torch.empty(n, device=pred.device, dtype=torch.float32)
    _check(lib().nlt_l2_loss_forward(_ptr(_dense(pred, 'pred')), _ptr(_dense(gt, 'gt')), n, pred[0].numel(),
                                     _ptr(loss), _stream()), 'nlt_l2_loss_forward')
    return loss


def l2_loss_backward(pred, gt, gloss):
    dpred = torch.empty_like(pred)
    _check(lib().nlt_l2_loss_backward(_ptr(pred), _ptr(gt), _ptr(_dense(gloss, 'gloss')), pred.shape[0],
                                      pred[0].numel(), _ptr(dpred), _stream()), 'nlt_l2_loss_backward')
    return dpred


def barron_loss(pred, gt, want_grad):
    n, h, w, c = pred.shape
    assert c == 3
    nws = lib().nlt_barron_workspace_floats(n, h, w)
    if nws <= 0:
        raise NLTError("nlt_barron_workspace_floats(%d,%d,%d) failed" % (n, h, w))
    ws = torch.empty(nws, device=pred.device, dtype=torch.float32)
    loss = torch.empty(n, device=pred.device, dtype=torch.float32)
    dunit = torch.empty_like(pred) if want_grad else None
    _check(lib().nlt_barron_loss(_ptr(_dense(pred, 'pred')), _ptr(_dense(gt, 'gt')), n, h, w, _ptr(ws), _ptr(loss),
                                 _ptr(dunit), _stream()), 'nlt_barron_loss')
    return loss, dunit


def scale_rows(x, scale):
    out = torch.empty_like(x)
    _check(lib().nlt_scale_rows(_ptr(_dense(x, 'x')), _ptr(_dense(scale, 'scale')), x.shape[0], x[0].numel(),
                                _ptr(out), _stream()), 'nlt_scale_rows')
    return out


def adam_amsgrad_step(param, grad, m, v, vhat, lr_t, beta1, beta2, eps):
    _check(lib().nlt_adam_amsgrad_step(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), _ptr(vhat), param.numel(),
                                       float(lr_t), float(beta1), float(beta2), float(eps), _stream()),
           'nlt_adam_amsgrad_step')
