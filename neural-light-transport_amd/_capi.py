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
