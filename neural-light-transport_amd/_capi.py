"""ctypes binding of libnlt_hip.so (include/nlt_hip.h) + thin torch-tensor adapters.

PyTorch is plumbing here (device memory + streams): every function takes torch CUDA tensors,
checks dtype/contiguity, and hands raw device pointers and the current HIP stream to the
C ABI.  There is NO CPU or torch fallback: if the library is missing or a call returns a
non-zero status this raises.
"""
import ctypes
import threading
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NLT_HIP_LIB') or os.path.join(_HERE, 'libnlt_hip.so')   # NLT_HIP_LIB: A/B of two builds (tools/ab_front.py)

CONV1X1, CONV_K2S2, CONV_K2S1, DECONV_K2S2, DECONV_K2S1 = range(5)
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA = range(3)

_c_int, _c_long, _c_float, _vp = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p
_c_double = ctypes.c_double
MAP_F64, MAP_F32, MAP_F16 = range(3)

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
    'nlt_warp_forward_store': (_c_int, [_vp] * 4 + [_c_int] * 5 + [_vp] * 5),
    'nlt_resample_forward': (_c_int, [_vp, _vp] + [_c_int] * 6 + [_vp, _vp]),
    'nlt_resize_bilinear_forward': (_c_int, [_vp] + [_c_int] * 6 + [_vp, _vp]),
    'nlt_mul_forward': (_c_int, [_vp, _vp, _c_long, _vp, _vp]),
    'nlt_conv_backward_weights': (_c_int, [_c_int, _c_int, _vp, _c_int, _c_int, _vp, _c_int, _c_int,
                                           _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _vp]),
    'nlt_wgrad_workspace_floats': (_c_long, [_c_int] * 7),
    'nlt_conv_backward_weights_tiled': (_c_int, [_c_int, _vp, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int,
                                                 _vp, _c_int, _c_int, _vp, _vp, _vp, _c_long, _vp]),
    'nlt_repack_weights': (_c_int, [_vp, _c_int, _c_long, _vp]),
    'nlt_tape_play': (_c_int, [_vp, _c_int, _vp]),
    'nlt_event_create': (_c_int, [_c_int, _vp]),
    'nlt_event_destroy': (_c_int, [_vp]),
    'nlt_event_record': (_c_int, [_vp, _vp]),
    'nlt_stream_wait_event': (_c_int, [_vp, _vp]),
    'nlt_wgrad_narrow_workspace_floats': (_c_long, [_c_int] * 7),
    'nlt_conv_backward_weights_narrow': (_c_int, [_c_int, _vp, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int,
                                                 _vp, _c_int, _c_int, _vp, _vp, _vp, _c_long, _vp]),
    'nlt_lrelu_backward': (_c_int, [_vp, _c_int, _vp, _c_int, _c_int, _c_long, _c_float, _vp, _c_int, _vp]),
    'nlt_obs_mean_backward': (_c_int, [_vp, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _vp, _vp]),
    'nlt_resize_cv_linear': (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_l2_train_loss': (_c_int, [_vp, _vp, _vp, _c_int, _c_long, _c_float, _vp, _vp, _vp, _vp]),
    'nlt_level_split_backward': (_c_int, [_vp, _vp, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _vp, _vp]),
    'nlt_stem_backward': (_c_int, [_vp] * 6 + [_c_int] * 5 + [_vp] * 6 + [_vp]),
    'nlt_head_backward': (_c_int, [_vp, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int,
                                   _vp, _c_int, _vp, _c_int, _vp, _vp, _vp]),
    'nlt_warp_backward': (_c_int, [_vp, _vp] + [_c_int] * 5 + [_vp, _vp]),
    'nlt_resize_bilinear_backward': (_c_int, [_vp] + [_c_int] * 6 + [_vp, _vp]),
    'nlt_l2_loss_forward': (_c_int, [_vp, _vp, _c_int, _c_long, _vp, _vp]),
    'nlt_l2_loss_backward': (_c_int, [_vp, _vp, _vp, _c_int, _c_long, _vp, _vp]),
    'nlt_l2_loss_weighted_forward': (_c_int, [_vp, _vp, _vp, _c_int, _c_long, _c_int, _vp, _vp]),
    'nlt_l2_loss_weighted_backward': (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_long, _c_int, _vp, _vp]),
    'nlt_barron_workspace_floats': (_c_long, [_c_int] * 3),
    'nlt_barron_loss': (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp]),
    'nlt_scale_rows': (_c_int, [_vp, _vp, _c_int, _c_long, _vp, _vp]),
    'nlt_sub_forward': (_c_int, [_vp, _vp, _c_long, _vp, _vp]),
    'nlt_finish_pred': (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_act_forward': (_c_int, [_vp, _c_long, _c_int, _c_float, _vp, _vp]),
    'nlt_act_backward': (_c_int, [_vp, _vp, _c_long, _c_int, _c_float, _vp, _vp]),
    'nlt_pixelnorm_forward': (_c_int, [_vp, _c_long, _c_int, _c_float, _vp, _vp]),
    'nlt_pixelnorm_backward': (_c_int, [_vp, _vp, _c_long, _c_int, _c_float, _vp, _vp]),
    'nlt_pool2x2_forward': (_c_int, [_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_pool2x2_backward': (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_norm_workspace_floats': (_c_long, [_c_long, _c_int]),
    'nlt_norm_forward': (_c_int, [_c_int, _vp, _c_long, _c_int, _vp, _vp, _vp, _vp, _c_float, _vp, _vp]),
    'nlt_norm_backward': (_c_int, [_c_int, _vp, _vp, _c_long, _c_int, _vp, _vp, _vp, _c_float, _vp, _vp, _vp, _vp, _vp]),
    'nlt_clip_by_norm_slots': (_c_int, [_vp, _vp, _c_int, _c_float, _vp]),
    'nlt_adam_amsgrad_step': (_c_int, [_vp] * 5 + [_c_long] + [_c_float] * 4 + [_vp]),
    'nlt_front_packed_floats': (_c_long, []),
    'nlt_front_pack_weights': (_c_int, [_vp] * 15 + [_vp]),
    'nlt_front_forward': (_c_int, [_vp] * 5 + [_c_int] * 4 + [_vp, _c_int, _c_float, _vp, _vp, _vp, _vp]),
    'nlt_front_l2_packed_floats': (_c_long, []),
    'nlt_front_pack_l2_weights': (_c_int, [_vp] * 5 + [_vp]),
    'nlt_front2_forward': (_c_int, [_vp] * 5 + [_c_int] * 4 + [_vp, _vp, _c_int, _c_float, _vp, _vp, _vp, _vp, _vp]),
    'nlt_front4_forward': (_c_int, [_vp] * 5 + [_c_int] * 4 + [_vp, _vp, _c_int, _c_float, _vp, _vp, _vp, _vp, _c_int, _vp]),
    'nlt_front4_forward_u8': (_c_int, [_vp] * 6 + [_c_int] * 4 + [_vp, _vp, _c_int, _c_float, _vp, _vp, _vp, _vp, _c_int, _vp]),
    'nlt_front4_forward_train': (_c_int, [_vp] * 5 + [_c_int] * 4 + [_vp, _vp, _c_int, _c_float] + [_vp] * 7 + [_vp]),
    'nlt_dec_block_forward': (_c_int, [_vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _c_float, _vp, _vp]),
    'nlt_back_forward': (_c_int, [_vp] * 3 + [_c_int] * 3 + [_vp] * 5 + [_c_float, _vp, _vp]),
    'nlt_front_forward_train': (_c_int, [_vp] * 5 + [_c_int] * 4 + [_vp, _c_int, _c_float] + [_vp] * 6),
    'nlt_back_forward_train': (_c_int, [_vp] * 3 + [_c_int] * 3 + [_vp] * 5 + [_c_float] + [_vp] * 4),
    'nlt_front_backward_workspace_floats': (_c_long, [_c_int] * 3),
    'nlt_front_backward': (_c_int, [_vp] * 5 + [_c_int] * 4 + [_vp] * 21),
    'nlt_back_backward_workspace_floats': (_c_long, [_c_int] * 3),
    'nlt_back_backward': (_c_int, [_vp] * 5 + [_c_int] * 3 + [_vp] * 3 + [_c_float] + [_vp] * 10),
    'nlt_conv_splitk_workspace_floats': (_c_long, [_c_int] * 6),
    'nlt_conv_forward_splitk': (_c_int, [_c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _vp, _c_int, _c_int,
                                         _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _c_int, _c_int, _c_float,
                                         _vp, _c_int, _c_int, _vp]),
    'nlt_conv_forward_map': (_c_int, [_c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _vp, _c_int, _c_int,
                                      _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _c_int, _c_int, _c_float,
                                      _vp, _c_int, _vp]),
    'nlt_dec_block_forward_map': (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_float, _vp, _vp, _vp]),
    'nlt_back_forward_map': (_c_int, [_vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_float, _vp, _vp, _vp]),
    'nlt_front_ovr_forward': (_c_int, [_vp] * 3 + [_c_int] * 3 + [_vp] * 5 + [_c_int, _c_float, _vp, _c_int, _vp, _vp, _vp]),
    'nlt_front_ovr_forward_u8': (_c_int, [_vp] * 4 + [_c_int] * 3 + [_vp] * 5 + [_c_int, _c_float, _vp, _c_int, _vp, _vp, _vp]),
    'nlt_conv_backward_data': (_c_int, [_c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int,
                                        _vp, _c_int, _vp, _c_int, _c_float, _c_int, _c_int, _vp, _vp, _c_float, _c_int, _vp]),
    'nlt_conv_tile_packed_floats': (_c_long, [_c_int] * 4),
    'nlt_pack_conv_tile_weights': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_conv_tile_forward': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int,
                                       _vp, _c_int, _vp, _c_int, _c_int, _c_float, _vp]),
    'nlt_pack_conv_tile_weights_adjoint': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_conv_c32_supported': (_c_int, [_c_int] * 3),
    'nlt_conv_c32_forward': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int,
                                      _vp, _c_int, _vp, _c_int, _c_int, _c_float, _vp]),
    'nlt_conv_wino_packed_floats': (_c_long, [_c_int] * 4),
    'nlt_pack_conv_wino_weights': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_conv_wino_forward': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int,
                                       _vp, _c_int, _vp, _c_int, _c_int, _c_float, _vp]),
    'nlt_pack_conv_wino_weights_adjoint': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_conv_wino_backward_data': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _c_int,
                                             _vp, _c_int, _c_float, _c_int, _vp]),
    'nlt_conv_tile_backward_data': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _c_int,
                                             _vp, _c_int, _c_float, _c_int, _c_int, _vp, _vp, _c_float, _c_int, _vp]),
    'nlt_conv_tile3_packed_elems': (_c_long, [_c_int] * 4),
    'nlt_pack_conv_tile3_weights': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_conv_tile3_forward': (_c_int, [_c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int,
                                        _vp, _c_int, _vp, _c_int, _c_int, _c_float, _vp]),
    'nlt_conv_bf16_packed_elems': (_c_long, [_c_int] * 4),
    'nlt_conv_bf16_pack': (_c_int, [_c_int, _vp, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_conv_bf16_forward': (_c_int, [_c_int, _c_int, _vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int,
                                       _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_float, _vp]),
    'nlt_obs_mean_bf16': (_c_int, [_vp, _c_int, _c_int, _c_long, _c_int, _vp, _c_int, _vp]),
    'nlt_chmix_bf16_packed_elems': (_c_long, [_c_int, _c_int]),
    'nlt_chmix_bf16_pack': (_c_int, [_vp, _c_int, _c_int, _vp, _vp]),
    'nlt_chmix_bf16_forward': (_c_int, [_vp, _c_long, _c_int, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp]),
    'nlt_cosine_map': (_c_int, [_vp] * 4 + [_c_double] * 3 + [_c_long, _vp, _vp, _vp]),
    'nlt_albedo': (_c_int, [_vp, _c_int, _c_long, _vp, _vp, _vp]),
    'nlt_diffuse_base': (_c_int, [_vp, _vp, _c_int, _c_long, _vp, _vp]),
    'nlt_remap_bilinear_u8': (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_remap_bilinear_f32': (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
    'nlt_uv_index_map_workspace_bytes': (_c_long, [_c_int, _c_int, _c_long]),
    'nlt_uv_index_map': (_c_int, [_vp, _vp, _c_long, _c_int, _c_int, _c_int, _c_int, _c_double, _vp, _vp, _vp, _vp]),
    'nlt_knn_indices': (_c_int, [_vp, _c_int, _vp, _c_int, _c_int, _vp, _vp]),
    'nlt_psnr_sums': (_c_int, [_vp, _vp, _vp, _c_long, _c_int, _vp, _vp, _vp]),
    'nlt_gather_frames_u8': (_c_int, [_vp, _vp, _c_int, _c_long, _vp, _vp]),
    'nlt_assemble_batch': (_c_int, [_vp] * 6 + [_c_int, _c_int, _c_long, _c_int] + [_vp] * 6 + [_vp]),
}

_real = None


def _load():
    """Loads libnlt_hip.so once; raises (loudly) when it has not been built."""
    global _real
    if _real is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libnlt_hip.so not found at %s -- build it with `python __graft_entry__.py` or "
                "`make -C neural-light-transport_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _real = L
    return _real


# ---------------------------------------------------------------- launch tape
# A plan (engine.RenderPlan) issues the same ~40 (forward) / ~130 (backward) C calls with the same arguments every
# step: same buffers, same weights, same streams.  Deriving those arguments again in Python (tensor -> pointer, layout
# checks, fragment-cache lookups, .detach() views) costs ~15-20 us per launch -- as much wall time as the GPU needs for
# the whole step at the training shape.  While a tape is open every launching C call is ALSO appended to it as
# (function, resolved arguments); a later step with the same inputs replays the list with nothing but the ctypes
# calls.  (hipGraph replay of the same sequence measured slower than eager launches on ROCm 7.0; see DESIGN.md.)
# The open tape belongs to the THREAD that opened it (pipeline.RenderPipeline drives one lane per host thread: a lane that
# records must not collect another lane's launches).
_tls = threading.local()
_alloc_epoch = [0]          # bumped whenever a cached device buffer the C calls point into is re-allocated



# int-returning entry points that answer a question instead of launching: a 0 from them is "no", not "launched"
_NOT_LAUNCHES = frozenset(('nlt_conv_c32_supported',))


class _TapeLib:
    """Stand-in for the CDLL while a tape is open: launching entry points (int status) are recorded after they ran."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if SIGNATURES[name][0] is not _c_int or name in _NOT_LAUNCHES:
            return fn                                   # size / capability queries: pure, not part of the step

        def recorded(*args):
            rc = fn(*args)
            t = getattr(_tls, 'tape', None)
            if t is not None and rc == 0:
                t.append((fn, args))
            return rc
        setattr(self, name, recorded)
        return recorded


_tape_lib = None


def lib():
    global _tape_lib
    if getattr(_tls, 'tape', None) is None:
        return _real if _real is not None else _load()
    if _tape_lib is None:
        _tape_lib = _TapeLib(_load())
    return _tape_lib


def tape_begin():
    if getattr(_tls, 'tape', None) is not None:
        raise NLTError("a launch tape is already open")
    _tls.tape = []
    _tls.epoch = _alloc_epoch[0]


def tape_end(tag=None, extra=None):
    """Closes the tape and returns [launch list, allocation epoch it is valid for, caller's validity tag, native form, extra]
    (`extra`: whatever the recorder wants back at replay time -- engine.py keeps the packed-buffer keys the tape reads)."""
    t, _tls.tape = _tls.tape, None
    if _tls.epoch != _alloc_epoch[0]:
        return None                                     # a cached buffer was re-allocated while recording: pointers are stale
    return (t, _alloc_epoch[0], tag, [None], extra)


def tape_abort():
    _tls.tape = None


def tape_pause():
    """Takes the open tape (if any) away for a launch that must NOT be recorded (its arguments change from step to step);
    hand the return value back to `tape_resume`."""
    t = getattr(_tls, 'tape', None)
    _tls.tape = None
    return t


def tape_resume(t):
    _tls.tape = t


def tape_valid(tape, tag=None):
    return tape is not None and tape[1] == _alloc_epoch[0] and tape[2] == tag


# Native replay (csrc/tape.hip: nlt_tape_play): the recorded calls of a tape, compiled once into arrays of nlt_tape_call and walked
# by ONE C call per run of consecutive native entries; what cannot be expressed (a Python hook, an entry with double arguments)
# stays a Python step between two runs.  OPT-IN (NLT_NATIVE_REPLAY=1): measured on MI355X boxes it changes nothing -- host enqueue
# 1.55 -> 1.52 ms per train step at config 4, 1.51 -> 1.48 at the 512^2 shape (r03): the ~9 us per launch are the HIP runtime's own
# (launch + the second-stream event traffic), not ctypes', and the steps are not host-bound to begin with.
NATIVE_REPLAY = os.environ.get('NLT_NATIVE_REPLAY', '0') != '0'


class _TapeCall(ctypes.Structure):
    _fields_ = [('fn', ctypes.c_void_p), ('n_float', ctypes.c_int), ('reserved', ctypes.c_int), ('iargs', ctypes.c_long * 32),
                ('fargs', ctypes.c_float * 4)]


def _describe(fn, args):
    """(address, integer-class args, float args) of one recorded step, or None when it has to stay a Python call."""
    owner = getattr(fn, '__self__', None)
    if owner is not None:                               # bound methods recorded by record_event / wait_event
        name = getattr(fn, '__name__', '')
        if isinstance(owner, torch.cuda.Event) and name == 'record' and len(args) == 1:
            return (ctypes.cast(_real.nlt_event_record, ctypes.c_void_p).value, [owner.cuda_event, args[0].cuda_stream], [])
        if isinstance(owner, torch.cuda.Stream) and name == 'wait_event' and len(args) == 1:
            return (ctypes.cast(_real.nlt_stream_wait_event, ctypes.c_void_p).value, [owner.cuda_stream, args[0].cuda_event], [])
        return None
    at = getattr(fn, 'argtypes', None)
    if at is None or len(at) != len(args):
        return None
    ia, fa = [], []
    for t, a in zip(at, args):
        if t is _c_float:
            fa.append(float(a))
        elif t in (_c_int, _c_long, _vp):
            ia.append(0 if a is None else int(a))
        else:
            return None                                 # doubles, char pointers: not a launching entry of a plan
    if len(ia) > 32 or len(fa) > 4:
        return None
    return (ctypes.cast(fn, ctypes.c_void_p).value, ia, fa)


def _compile(calls):
    """[('native', array, count) | ('py', fn, args)] in order."""
    segs, run = [], []

    def flush():
        if run:
            arr = (_TapeCall * len(run))()
            for i, (addr, ia, fa) in enumerate(run):
                arr[i].fn, arr[i].n_float = addr, len(fa)
                for j, v in enumerate(ia):
                    arr[i].iargs[j] = v
                for j, v in enumerate(fa):
                    arr[i].fargs[j] = v
            segs.append(('native', arr, len(run)))
            del run[:]
    for fn, args in calls:
        d = _describe(fn, args)
        if d is None or not d[0]:
            flush()
            segs.append(('py', fn, args))
        else:
            run.append(d)
    flush()
    return segs


def replay(tape):
    if (NATIVE_REPLAY or getattr(_tls, 'native_replay', False)) and len(tape) > 3 and _real is not None:
        box = tape[3]
        if box[0] is None:
            box[0] = _compile(tape[0])
        failed = ctypes.c_int(-1)
        for seg in box[0]:
            if seg[0] == 'native':
                rc = _real.nlt_tape_play(seg[1], seg[2], ctypes.byref(failed))
                if rc:
                    raise NLTError("replayed launch %d failed: %s" % (failed.value, _real.nlt_status_string(rc).decode()))
            else:
                rc = seg[1](*seg[2])
                if rc:
                    raise NLTError("replayed launch failed: %s" % lib().nlt_status_string(rc).decode())
        return
    for fn, args in tape[0]:
        rc = fn(*args)
        if rc:                                          # event helpers return None
            raise NLTError("replayed launch failed: %s" % lib().nlt_status_string(rc).decode())


def tape_call(fn, *args):
    """A host-side step that belongs to the plan (a hook): run it now and, while a tape is open, on every replay."""
    fn(*args)
    t = getattr(_tls, 'tape', None)
    if t is not None:
        t.append((fn, args))


class LightEvent:
    """A HIP event for ordering two streams of one device: no timing, no system-scope fence at the record (csrc/tape.hip).  Record
    and wait go through the C ABI, so an open launch tape picks them up like any launch (and replays them natively)."""

    def __init__(self):
        h = ctypes.c_void_p()
        _check(_load().nlt_event_create(1, ctypes.byref(h)), 'nlt_event_create')
        self.handle = h.value

    def __del__(self):
        try:
            if self.handle and _real is not None:
                _real.nlt_event_destroy(self.handle)
        except Exception:
            pass


LIGHT_EVENTS = None         # None: decide per event (below); True / False: forced (tests, the bench's A/B leg)


def light_events_enabled():
    """Events WITHOUT a system-scope fence at the record (csrc/tape.hip) order the plan's device-to-device hand-overs; the
    kernels' own agent-scope release / acquire orders the data.  In a multi-rank process group one of those hand-overs feeds the
    side-stream RCCL all-reduce of a gradient range, whose peers read the bucket over xGMI: until that has been A/B-tested
    on real multi-GPU hardware (bench.py's `light_events_ab` leg does it, bitwise) the default at world > 1 is the FENCED event
    (advisor r04 / review r05).  NLT_LIGHT_EVENTS=0 / 1 forces either."""
    if LIGHT_EVENTS is not None:
        return bool(LIGHT_EVENTS)
    e = os.environ.get('NLT_LIGHT_EVENTS')
    if e is not None:
        return e != '0'
    try:
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
    except Exception:
        return True


def new_event():
    """Event for the plan's cross-stream hand-overs (`light_events_enabled`; otherwise an ordinary torch.cuda.Event)."""
    return LightEvent() if light_events_enabled() else torch.cuda.Event()


def record_event(ev, stream):
    if isinstance(ev, LightEvent):
        _check(lib().nlt_event_record(ev.handle, stream.cuda_stream), 'nlt_event_record')
        return
    ev.record(stream)
    t = getattr(_tls, 'tape', None)
    if t is not None:
        t.append((ev.record, (stream,)))


def wait_event(stream, ev):
    if isinstance(ev, LightEvent):
        _check(lib().nlt_stream_wait_event(stream.cuda_stream, ev.handle), 'nlt_stream_wait_event')
        return
    stream.wait_event(ev)
    t = getattr(_tls, 'tape', None)
    if t is not None:
        t.append((stream.wait_event, (ev,)))


_WS_SCOPE = os.environ.get('NLT_WS_SCOPE', '1') != '0'      # (A/B switch)


def set_workspace_scope(token):
    """Scratch that is cached per stream (the split-K partial sums) is additionally keyed by this token; a plan sets it to its
    own identity on entry.  Two plans whose launches are CAPTURED on the same capture stream (pipeline.RenderPipeline lanes
    replaying hipGraphs) would otherwise bake one workspace into both graphs and race on it when the graphs replay side by side."""
    _tls.scope = token if _WS_SCOPE else 0


def drop_workspace_scope(token):
    """Forgets the scratch cached for a plan that is gone (RenderPlan.__del__: the lanes of a closed pipeline)."""
    for key in [k for k in _splitk_ws if len(k) > 2 and k[2] == token]:
        del _splitk_ws[key]


def set_thread_native_replay(on):
    """This host thread replays its launch tapes through nlt_tape_play (one C call per run of launches, the interpreter lock
    released for all of it): what lets the lanes of pipeline.RenderPipeline enqueue in parallel."""
    _tls.native_replay = bool(on)


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


def _tptr(t, dtype, what):
    """Device pointer of a contiguous CUDA tensor of exactly `dtype` (None passes through)."""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise NLTError("%s: expected a contiguous %s CUDA tensor, got %s on %s" % (what, dtype, t.dtype, t.device))
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


def warp_forward_store(pred, diffuse_store, uv2cam_store, ids, n, uvh, uvw, hc, wc, pred_cam, base_cam, fg_cam, idx_out=None):
    """warp_forward for a store-resident batch: base / map read from the uint8 diffuse and fp16 uv2cam stores (frame ids)."""
    if tuple(diffuse_store.shape[1:]) != (uvh, uvw, 3) or tuple(uv2cam_store.shape[1:]) != (hc, wc, 2):
        raise NLTError("stores %s / %s do not match uv %dx%d, camera %dx%d"
                       % (tuple(diffuse_store.shape), tuple(uv2cam_store.shape), uvh, uvw, hc, wc))
    _check(lib().nlt_warp_forward_store(_ptr(pred), _tptr(diffuse_store, torch.uint8, 'diffuse store'),
                                        _tptr(uv2cam_store, torch.float16, 'uv2cam store'), _tptr(ids, torch.int32, 'ids'),
                                        n, uvh, uvw, hc, wc, _ptr(pred_cam), _ptr(base_cam), _ptr(fg_cam), _ptr(idx_out),
                                        _stream()), 'nlt_warp_forward_store')


def resample_forward(data, warp_px):
    """tfa.image.resampler(data [n,h,w,c], warp_px [n,hc,wc,2] in pixel units) for c % 4 == 0 (the 64-channel stress point)."""
    n, h, w, c = data.shape
    hc, wc = warp_px.shape[1:3]
    out = torch.empty((n, hc, wc, c), device=data.device, dtype=torch.float32)
    _check(lib().nlt_resample_forward(_ptr(_dense(data, 'data')), _ptr(_dense(warp_px, 'warp_px')), n, h, w, c, hc, wc, _ptr(out),
                                      _stream()), 'nlt_resample_forward')
    return out


def resize_bilinear_forward(x, oh, ow):
    n, h, w, c = x.shape
    out = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
    _check(lib().nlt_resize_bilinear_forward(_ptr(_dense(x, 'x')), n, h, w, c, oh, ow, _ptr(out), _stream()),
           'nlt_resize_bilinear_forward')
    return out


def _same_shape(a, b, what):
    if tuple(a.shape) != tuple(b.shape):
        raise NLTError("%s: shapes differ, %s vs %s" % (what, tuple(a.shape), tuple(b.shape)))


def mul_forward(a, b):
    _same_shape(a, b, 'mul_forward')
    out = torch.empty_like(a)
    _check(lib().nlt_mul_forward(_ptr(_dense(a, 'a')), _ptr(_dense(b, 'b')), a.numel(), _ptr(out), _stream()),
           'nlt_mul_forward')
    return out


# ---------------------------------------------------------------- train step
def conv_backward_weights(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp, cout, dw, db, algo=ALGO_AUTO):
    _check(lib().nlt_conv_backward_weights(mode, algo, _ptr(src0), ld0, c0, _ptr(src1), ld1, c1, n, h, w,
                                           _ptr(dpre), ldp, cout, _ptr(dw), _ptr(db), _stream()),
           'nlt_conv_backward_weights')


_wgrad_ws = {}
_named_ws = {}


def _workspace(name, device, need):
    """Scratch floats for the two-pass (deterministic) reductions, cached per (kernel family, device), grown on demand."""
    key = (name, str(device))
    ws = _named_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=device, dtype=torch.float32)
        _named_ws[key] = ws
        _alloc_epoch[0] += 1                         # recorded launch tapes point into the old buffer
    return ws


def conv_backward_weights_tiled(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp, cout, dw, db):
    """Second-generation weight gradient (deterministic two-pass); the slice workspace is cached per device."""
    need = lib().nlt_wgrad_workspace_floats(mode, c0, c1, n, h, w, cout)
    if need <= 0:
        raise NLTError("nlt_wgrad_workspace_floats: unsupported (mode %d, c0 %d, c1 %d, cout %d)" % (mode, c0, c1, cout))
    key = (str(src0.device), _stream())                 # per stream: the plan may deal weight gradients to two streams
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=src0.device, dtype=torch.float32)
        _wgrad_ws[key] = ws
        _alloc_epoch[0] += 1
    _check(lib().nlt_conv_backward_weights_tiled(mode, _ptr(src0), ld0, c0, _ptr(src1), ld1, c1, n, h, w, _ptr(dpre), ldp,
                                                 cout, _ptr(dw), _ptr(db), _ptr(ws), ws.numel(), _stream()),
           'nlt_conv_backward_weights_tiled')


def wgrad_narrow_supported(mode, c0, c1, n, h, w, cout):
    return lib().nlt_wgrad_narrow_workspace_floats(mode, c0, c1, n, h, w, cout) > 0


def conv_backward_weights_narrow(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp, cout, dw, db):
    """Weight gradient of a narrow layer (<= 32 output columns, K <= 128): MFMA tile matched to the layer."""
    need = lib().nlt_wgrad_narrow_workspace_floats(mode, c0, c1, n, h, w, cout)
    if need <= 0:
        raise NLTError("nlt_wgrad_narrow_workspace_floats: unsupported (mode %d, c0 %d, c1 %d, cout %d)" % (mode, c0, c1, cout))
    ws = _workspace('wgrad_narrow.%d' % _stream(), src0.device, need)
    _check(lib().nlt_conv_backward_weights_narrow(mode, _ptr(src0), ld0, c0, _ptr(src1), ld1, c1, n, h, w, _ptr(dpre), ldp,
                                                  cout, _ptr(dw), _ptr(db), _ptr(ws), ws.numel(), _stream()),
           'nlt_conv_backward_weights_narrow')


def lrelu_backward(g, ldg, y, ldy, c, texels, alpha, out, ldo):
    _check(lib().nlt_lrelu_backward(_ptr(g), ldg, _ptr(y), ldy, c, texels, float(alpha), _ptr(out), ldo, _stream()),
           'nlt_lrelu_backward')


def obs_mean_backward(dmean, ldm, obs_y, obs_weights, dobs_partial, n, k, hw, c, alpha, dpre_obs):
    _check(lib().nlt_obs_mean_backward(_ptr(dmean), ldm, _ptr(obs_y), _ptr(obs_weights), _ptr(dobs_partial),
                                       n, k, hw, c, float(alpha), _ptr(dpre_obs), _stream()), 'nlt_obs_mean_backward')


def resize_cv_linear(src, oh, ow, out=None):
    """cv2.resize(normalised src, (ow, oh)) (INTER_LINEAR) -> float32.  src [n,h,w,c] uint8, int32 (16-bit samples) or
    float32 (already normalised)."""
    kind = {torch.uint8: 0, torch.int32: 1, torch.float32: 2}.get(src.dtype)
    if kind is None or src.dim() != 4:
        raise NLTError("resize_cv_linear: [n,h,w,c] uint8 / int32 / float32 expected, got %s %s" % (src.dtype, tuple(src.shape)))
    src = _dense(src, 'src')
    n, h, w, c = src.shape
    if out is None:
        out = torch.empty((n, oh, ow, c), device=src.device, dtype=torch.float32)
    _check(lib().nlt_resize_cv_linear(_tptr(src, src.dtype, 'src'), kind, n, h, w, c, oh, ow, _ptr(out), _stream()), 'nlt_resize_cv_linear')
    return out


def level_split_backward(dfm, fm_y, ld, obs_y, obs_weights, dobs_partial, n, k, hw, c, alpha_q, alpha_o, dpre_obs):
    """lrelu_backward on the query half (in place) + obs_mean_backward on the observation half of dfm [n,hw,ld], one launch."""
    _check(lib().nlt_level_split_backward(_ptr(dfm), _ptr(fm_y), ld, _ptr(obs_y), _ptr(obs_weights), _ptr(dobs_partial),
                                          n, k, hw, c, float(alpha_q), float(alpha_o), _ptr(dpre_obs), _stream()),
           'nlt_level_split_backward')


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
    _same_shape(pred, gt, 'l2_loss_forward')
    n = pred.shape[0]
    loss = torch.empty(n, device=pred.device, dtype=torch.float32)
    _check(lib().nlt_l2_loss_forward(_ptr(_dense(pred, 'pred')), _ptr(_dense(gt, 'gt')), n, pred[0].numel(),
                                     _ptr(loss), _stream()), 'nlt_l2_loss_forward')
    return loss


def l2_train_loss(pred, rgb, fg, global_bs):
    """gt = rgb * fg, loss = sum_f mean((pred_f - gt_f)^2) / global_bs (0-dim tensor), dpred: one launch."""
    _same_shape(pred, rgb, 'l2_train_loss'); _same_shape(pred, fg, 'l2_train_loss')
    gt, dpred = torch.empty_like(pred), torch.empty_like(pred)
    loss = torch.empty((), device=pred.device, dtype=torch.float32)
    _check(lib().nlt_l2_train_loss(_ptr(_dense(pred, 'pred')), _ptr(_dense(rgb, 'rgb')), _ptr(_dense(fg, 'fg')), pred.shape[0],
                                   pred[0].numel(), 1.0 / float(global_bs), _ptr(gt), _ptr(dpred), _ptr(loss), _stream()),
           'nlt_l2_train_loss')
    return loss, gt, dpred


def l2_loss_backward(pred, gt, gloss):
    _same_shape(pred, gt, 'l2_loss_backward')
    if gloss.numel() != pred.shape[0]:
        raise NLTError("l2_loss_backward: %d loss gradients for %d examples" % (gloss.numel(), pred.shape[0]))
    dpred = torch.empty_like(pred)
    _check(lib().nlt_l2_loss_backward(_ptr(pred), _ptr(gt), _ptr(_dense(gloss, 'gloss')), pred.shape[0],
                                      pred[0].numel(), _ptr(dpred), _stream()), 'nlt_l2_loss_backward')
    return dpred


def l2_loss_weighted_forward(pred, gt, weights):
    """weights [N,H,W] (already broadcast): loss[f] = mean_hw(weights * mean_c (gt - pred)^2)."""
    _same_shape(pred, gt, 'l2_loss_weighted_forward')
    n, c = pred.shape[0], pred.shape[-1]
    hw = pred[0].numel() // c
    if weights.numel() != n * hw:
        raise NLTError("l2_loss_weighted_forward: %d weights for %d texels" % (weights.numel(), n * hw))
    loss = torch.empty(n, device=pred.device, dtype=torch.float32)
    _check(lib().nlt_l2_loss_weighted_forward(_ptr(_dense(pred, 'pred')), _ptr(_dense(gt, 'gt')), _ptr(_dense(weights, 'weights')),
                                              n, hw, c, _ptr(loss), _stream()), 'nlt_l2_loss_weighted_forward')
    return loss


def l2_loss_weighted_backward(pred, gt, weights, gloss):
    _same_shape(pred, gt, 'l2_loss_weighted_backward')
    n, c = pred.shape[0], pred.shape[-1]
    hw = pred[0].numel() // c
    if gloss.numel() != n or weights.numel() != n * hw:
        raise NLTError("l2_loss_weighted_backward: %d loss gradients, %d weights for %d examples of %d texels"
                       % (gloss.numel(), weights.numel(), n, hw))
    dpred = torch.empty_like(pred)
    _check(lib().nlt_l2_loss_weighted_backward(_ptr(_dense(pred, 'pred')), _ptr(_dense(gt, 'gt')), _ptr(_dense(weights, 'weights')),
                                               _ptr(_dense(gloss, 'gloss')), n, hw, c, _ptr(dpred), _stream()),
           'nlt_l2_loss_weighted_backward')
    return dpred


def barron_loss(pred, gt, want_grad):
    _same_shape(pred, gt, 'barron_loss')
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
    if scale.numel() != x.shape[0]:
        raise NLTError("scale_rows: %d scales for %d rows" % (scale.numel(), x.shape[0]))
    out = torch.empty_like(x)
    _check(lib().nlt_scale_rows(_ptr(_dense(x, 'x')), _ptr(_dense(scale, 'scale')), x.shape[0], x[0].numel(),
                                _ptr(out), _stream()), 'nlt_scale_rows')
    return out


# ---------------------------------------------------------------- layers of the non-default config branches
ACT_LRELU, ACT_ELU = 0, 1
POOL_MAX, POOL_AVG = 0, 1


def sub_forward(a, b):
    _same_shape(a, b, 'sub_forward')
    out = torch.empty_like(a)
    _check(lib().nlt_sub_forward(_ptr(_dense(a, 'a')), _ptr(_dense(b, 'b')), a.numel(), _ptr(out), _stream()), 'nlt_sub_forward')
    return out


def finish_pred(y, base, pred):
    n, h, w, c = y.shape
    if c != 3 or (base is not None and tuple(base.shape) != tuple(y.shape)) or tuple(pred.shape) != tuple(y.shape):
        raise NLTError("finish_pred: y / base / pred must be [n,h,w,3] of one shape")
    _check(lib().nlt_finish_pred(_ptr(_dense(y, 'y')), _ptr(base), n, h, w, _ptr(pred), _stream()), 'nlt_finish_pred')


def act_forward(x, kind, alpha):
    y = torch.empty_like(x)
    _check(lib().nlt_act_forward(_ptr(_dense(x, 'x')), x.numel(), kind, float(alpha), _ptr(y), _stream()), 'nlt_act_forward')
    return y


def act_backward(g, y, kind, alpha):
    _same_shape(g, y, 'act_backward')
    dx = torch.empty_like(y)
    _check(lib().nlt_act_backward(_ptr(_dense(g, 'g')), _ptr(_dense(y, 'y')), y.numel(), kind, float(alpha), _ptr(dx), _stream()),
           'nlt_act_backward')
    return dx


def pixelnorm_forward(x, eps=1e-8):
    y = torch.empty_like(x)
    c = x.shape[-1]
    _check(lib().nlt_pixelnorm_forward(_ptr(_dense(x, 'x')), x.numel() // c, c, float(eps), _ptr(y), _stream()), 'nlt_pixelnorm_forward')
    return y


def pixelnorm_backward(g, x, eps=1e-8):
    _same_shape(g, x, 'pixelnorm_backward')
    dx = torch.empty_like(x)
    c = x.shape[-1]
    _check(lib().nlt_pixelnorm_backward(_ptr(_dense(g, 'g')), _ptr(_dense(x, 'x')), x.numel() // c, c, float(eps), _ptr(dx), _stream()),
           'nlt_pixelnorm_backward')
    return dx


NORM_LAYER, NORM_BATCH = 0, 1


def norm_forward(kind, x, gamma, beta, mean, var, eps):
    """LayerNormalization (kind 0) / inference-mode BatchNormalization (kind 1) over the channel axis of x [..., c]."""
    c = x.shape[-1]
    y = torch.empty_like(x)
    _check(lib().nlt_norm_forward(kind, _ptr(_dense(x, 'x')), x.numel() // c, c, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var),
                                  float(eps), _ptr(y), _stream()), 'nlt_norm_forward')
    return y


def norm_backward(kind, g, x, gamma, mean, var, eps, dgamma, dbeta):
    """dx; dgamma / dbeta are ACCUMULATED into the given views (of the flat gradient bucket)."""
    _same_shape(g, x, 'norm_backward')
    c = x.shape[-1]
    texels = x.numel() // c
    need = lib().nlt_norm_workspace_floats(texels, c)
    if need < 0:
        raise NLTError("norm over %d channels (the kernel takes c <= 1024)" % c)
    ws = _workspace('norm', x.device, need)
    dx = torch.empty_like(x)
    _check(lib().nlt_norm_backward(kind, _ptr(_dense(g, 'g')), _ptr(_dense(x, 'x')), texels, c, _ptr(gamma), _ptr(mean), _ptr(var),
                                   float(eps), _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), _stream()), 'nlt_norm_backward')
    return dx


def pool2x2_forward(x, kind):
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), device=x.device, dtype=torch.float32)
    _check(lib().nlt_pool2x2_forward(_ptr(_dense(x, 'x')), n, h, w, c, kind, _ptr(y), _stream()), 'nlt_pool2x2_forward')
    return y


def pool2x2_backward(g, x, kind):
    n, h, w, c = x.shape
    if tuple(g.shape) != (n, h // 2, w // 2, c):
        raise NLTError("pool2x2_backward: gradient %s for input %s" % (tuple(g.shape), tuple(x.shape)))
    dx = torch.empty_like(x)
    _check(lib().nlt_pool2x2_backward(_ptr(_dense(g, 'g')), _ptr(_dense(x, 'x')), n, h, w, c, kind, _ptr(dx), _stream()),
           'nlt_pool2x2_backward')
    return dx


def clip_by_norm_slots(grad, slots, clipnorm):
    """grad: flat fp32 bucket; slots: int64 [n,2] (offset, count) on the same device; in place."""
    _check(lib().nlt_clip_by_norm_slots(_ptr(grad), _tptr(slots, torch.int64, 'slots'), slots.shape[0], float(clipnorm),
                                        _stream()), 'nlt_clip_by_norm_slots')


def adam_amsgrad_step(param, grad, m, v, vhat, lr_t, beta1, beta2, eps):
    _check(lib().nlt_adam_amsgrad_step(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), _ptr(vhat), param.numel(),
                                       float(lr_t), float(beta1), float(beta2), float(eps), _stream()),
           'nlt_adam_amsgrad_step')


_splitk_ws = {}


def _splitk_workspace(need, device):
    """The split-K scratch of the current (device, stream, plan): cached, grown on demand, ZEROED when allocated -- its first words are
    the tiles' ticket counters, which every launch leaves at zero (include/nlt_hip.h: nlt_conv_splitk_workspace_floats)."""
    key = (str(device), _stream(), getattr(_tls, 'scope', 0))   # per stream (the query and observation paths run concurrently) and per plan (see set_workspace_scope)
    ws = _splitk_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.zeros(need, device=device, dtype=torch.float32)
        _splitk_ws[key] = ws
        _alloc_epoch[0] += 1
    return ws


def conv_forward_splitk(mode, ksplit, src0, c0, ld0, src1, c1, ld1, n, h, w, w_packed, bias, cout, out, ldo,
                        act=True, alpha=0.3, tile_hint=0, mask_src=None, ldm=0, accumulate=False):
    """nlt_conv_forward (MFMA path) with the K loop split over |ksplit| wave slices (ksplit > 0: one launch, the last workgroup
    to arrive at a tile adds the groups' partial tiles; ksplit < 0: the two-launch form of rounds 2-5); the partial-sum
    workspace is cached per device and grown on demand."""
    need = lib().nlt_conv_splitk_workspace_floats(mode, n, h, w, cout, ksplit)
    if need <= 0:
        raise NLTError("nlt_conv_splitk_workspace_floats failed")
    ws = _splitk_workspace(need, src0.device)
    _check(lib().nlt_conv_forward_splitk(mode, tile_hint, ksplit, _ptr(ws), _ptr(src0), ld0, c0, _ptr(src1), ld1, c1, n, h, w,
                                         _ptr(w_packed), _ptr(bias), cout, _ptr(out), ldo, 1 if act else 0, float(alpha),
                                         _ptr(mask_src), ldm, 1 if accumulate else 0, _stream()), 'nlt_conv_forward_splitk')


def conv_forward_map(mode, ksplit, src0, c0, ld0, src1, c1, ld1, n, h, w, w_packed, bias, cout, out, ldo, bias_map,
                     act=True, alpha=0.3, tile_hint=0, w_keras=None):
    """out = act(conv(src0 | src1) + bias + bias_map): the per-frame half of a conv over concat(q, given observation map)
    (include/nlt_hip.h: nlt_conv_forward_map).  bias_map [1 or n, oh, ow, cout] dense.  (w_keras is not read here; the host
    tests' CPU emulation of this adapter computes from it.)"""
    ws = None
    if abs(ksplit) > 1:
        need = lib().nlt_conv_splitk_workspace_floats(mode, n, h, w, cout, ksplit)
        if need <= 0:
            raise NLTError("nlt_conv_splitk_workspace_floats failed")
        ws = _splitk_workspace(need, src0.device)
    _check(lib().nlt_conv_forward_map(mode, tile_hint, ksplit, _ptr(ws), _ptr(src0), ld0, c0, _ptr(src1), ld1, c1, n, h, w,
                                      _ptr(w_packed), _ptr(bias), cout, _ptr(out), ldo, 1 if act else 0, float(alpha),
                                      _ptr(_dense(bias_map, 'bias_map')), bias_map.shape[0], _stream()), 'nlt_conv_forward_map')


def conv_backward_data(adj_mode, dpre, cpre, ldp, n, h, w, w_packed, zero_bias, cout, out, ldo, mask_src=None, ldm=0,
                       mask_alpha=0.3, accumulate=False, tile_hint=0, ksplit=1, split=None, w_keras=None):
    """Gradient w.r.t. a conv's input channels (include/nlt_hip.h: nlt_conv_backward_data).  split = (c, obs_y, dobs,
    alpha_o, has_partial): the target is dfm[l] with one observation per frame; its observation half goes, finished, to dobs.
    (w_keras -- the Keras-layout slice the fragments were packed from -- is not read here; the host tests' CPU emulation of
    this adapter computes from it.)"""
    ws = None
    if abs(ksplit) > 1:
        need = lib().nlt_conv_splitk_workspace_floats(adj_mode, n, h, w, cout, ksplit)
        if need <= 0:
            raise NLTError("nlt_conv_splitk_workspace_floats failed")
        ws = _splitk_workspace(need, dpre.device)
    sc, sy, sd, sa, sp = split if split is not None else (0, None, None, 0.0, False)
    _check(lib().nlt_conv_backward_data(adj_mode, tile_hint, ksplit, _ptr(ws), _ptr(dpre), ldp, cpre, n, h, w, _ptr(w_packed),
                                        _ptr(zero_bias), cout, _ptr(out), ldo, _ptr(mask_src), ldm, float(mask_alpha),
                                        1 if accumulate else 0, sc, _ptr(sy), _ptr(sd), float(sa), 1 if sp else 0, _stream()),
           'nlt_conv_backward_data')


# ---------------------------------------------------------------- one-launch refresh of all packed weights
REPACK_MFMA, REPACK_TILE, REPACK_WINO = 0, 1, 2
REPACK_FIELDS = [('src', 'u8'), ('dst', 'u8'), ('total', 'i8'), ('first_block', 'i8'), ('kind', 'i4'), ('mode', 'i4'),
                 ('c0', 'i4'), ('c1', 'i4'), ('cout', 'i4'), ('tn', 'i4'), ('lo', 'i4'), ('full', 'i4')]   # = nlt_repack_desc


def repack_table(entries, device):
    """entries: dicts with the nlt_repack_desc fields except first_block (src / dst as tensors).  Returns the device
    table (uint8 tensor), the number of descriptors and the grid size for nlt_repack_weights."""
    import numpy as np
    tab = np.zeros(len(entries), dtype=np.dtype(REPACK_FIELDS))
    blocks = 0
    for i, e in enumerate(entries):
        if e['dst'].numel() % 4 or e['dst'].data_ptr() % 16:          # (the refresh launch stores quads)
            raise NLTError("repack_table: a packed buffer must be 16-byte aligned and a multiple of 4 floats long")
        tab[i] = (e['src'].data_ptr(), e['dst'].data_ptr(), e['dst'].numel(), blocks, e['kind'], e['mode'], e['c0'], e['c1'],
                  e['cout'], e['tn'], e['lo'], e['full'])
        blocks += (e['dst'].numel() + 255) // 256
    return torch.from_numpy(tab.view(np.uint8)).to(device), len(entries), blocks


def repack_weights(table, n_desc, total_blocks):
    _check(lib().nlt_repack_weights(_tptr(table, torch.uint8, 'repack table'), n_desc, total_blocks, _stream()),
           'nlt_repack_weights')


# ---------------------------------------------------------------- LDS-tiled encoder convs
def conv_tile_supported(mode, cin, cout, tn):
    return lib().nlt_conv_tile_packed_floats(mode, cin, cout, tn) > 0


def pack_conv_tile_weights(mode, w_keras, cin, cout, tn):
    n = lib().nlt_conv_tile_packed_floats(mode, cin, cout, tn)
    if n <= 0:
        raise NLTError("conv_tile: unsupported (mode %d, cin %d, cout %d, tn %d)" % (mode, cin, cout, tn))
    out = torch.empty(n, device=w_keras.device, dtype=torch.float32)
    _check(lib().nlt_pack_conv_tile_weights(mode, _ptr(_dense(w_keras, 'w_keras')), cin, cout, tn, _ptr(out), _stream()),
           'nlt_pack_conv_tile_weights')
    return out


def conv_tile_forward(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, tn, out, ldo, mean_out, ldm,
                      act=True, alpha=0.3):
    _check(lib().nlt_conv_tile_forward(mode, _ptr(src), ld, cin, frames, kobs, h, w, _ptr(packed), _ptr(bias), cout, tn,
                                       _ptr(out), ldo, _ptr(mean_out), ldm, 1 if act else 0, float(alpha), _stream()),
           'nlt_conv_tile_forward')


def conv_c32_supported(mode, cin, cout):
    return lib().nlt_conv_c32_supported(mode, cin, cout) > 0


def conv_c32_forward(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, out, ldo, mean_out, ldm, act=True, alpha=0.3):
    """Narrow stride-1 conv (cin 16 | 32 -> 32) with LDS-resident weights; packed = pack_conv_tile_weights(mode, w, cin, 32, 32)."""
    _check(lib().nlt_conv_c32_forward(mode, _ptr(src), ld, cin, frames, kobs, h, w, _ptr(packed), _ptr(bias), cout,
                                      _ptr(out), ldo, _ptr(mean_out), ldm, 1 if act else 0, float(alpha), _stream()),
           'nlt_conv_c32_forward')


# ---------------------------------------------------------------- Winograd stride-1 k2 convs (csrc/conv_wino.hip)
def conv_wino_supported(mode, cin, cout, tn):
    return lib().nlt_conv_wino_packed_floats(mode, cin, cout, tn) > 0


def pack_conv_wino_weights(mode, w_keras, cin, cout, tn, full=None, lo=0):
    """G g G^T fragments of a stride-1 k2 kernel; full / lo: the ADJOINT family read from a layer's own array (columns = that
    layer's input channels [lo, lo + cout) of `full`)."""
    n = lib().nlt_conv_wino_packed_floats(mode, cin, cout, tn)
    if n <= 0:
        raise NLTError("conv_wino: unsupported (mode %d, cin %d, cout %d, tn %d)" % (mode, cin, cout, tn))
    out = torch.empty(n, device=w_keras.device, dtype=torch.float32)
    if full is None:
        _check(lib().nlt_pack_conv_wino_weights(mode, _ptr(_dense(w_keras, 'w_keras')), cin, cout, tn, _ptr(out), _stream()),
               'nlt_pack_conv_wino_weights')
    else:
        _check(lib().nlt_pack_conv_wino_weights_adjoint(mode, _ptr(_dense(w_keras, 'w_keras')), cin, cout, tn, full, lo, _ptr(out),
                                                        _stream()), 'nlt_pack_conv_wino_weights_adjoint')
    return out


def conv_wino_forward(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, tn, out, ldo, mean_out, ldm, act=True, alpha=0.3):
    _check(lib().nlt_conv_wino_forward(mode, _ptr(src), ld, cin, frames, kobs, h, w, _ptr(packed), _ptr(bias), cout, tn,
                                       _ptr(out), ldo, _ptr(mean_out), ldm, 1 if act else 0, float(alpha), _stream()),
           'nlt_conv_wino_forward')


def conv_wino_backward_data(adj_mode, dpre, cpre, ldp, n, h, w, packed, cout, tn, out, ldo, mask_src=None, ldm=0, mask_alpha=0.3,
                            accumulate=False):
    _check(lib().nlt_conv_wino_backward_data(adj_mode, _ptr(dpre), ldp, cpre, n, h, w, _ptr(packed), cout, tn, _ptr(out), ldo,
                                             _ptr(mask_src), ldm, float(mask_alpha), 1 if accumulate else 0, _stream()),
           'nlt_conv_wino_backward_data')


# ---------------------------------------------------------------- bf16 middle of the network
def pack_conv_tile_weights_adjoint(adj_mode, w_keras, cpre, cout, tn, full, lo):
    """Tile fragments of the ADJOINT conv family read from the layer's own (contiguous) Keras array: output columns = the layer's
    input channels [lo, lo + cout) of `full`."""
    n = lib().nlt_conv_tile_packed_floats(adj_mode, cpre, cout, tn)
    if n <= 0:
        raise NLTError("nlt_conv_tile_packed_floats: unsupported (mode %d, cpre %d, cout %d, tn %d)" % (adj_mode, cpre, cout, tn))
    out = torch.empty(n, device=w_keras.device, dtype=torch.float32)
    _check(lib().nlt_pack_conv_tile_weights_adjoint(adj_mode, _ptr(_dense(w_keras, 'w_keras')), cpre, cout, tn, full, lo, _ptr(out),
                                                    _stream()), 'nlt_pack_conv_tile_weights_adjoint')
    return out


def conv_tile_backward_data(adj_mode, dpre, cpre, ldp, n, h, w, packed, cout, tn, out, ldo, mask_src=None, ldm=0, mask_alpha=0.3,
                            accumulate=False, split=None, w_keras=None):
    """Backward-data on the LDS-tiled kernel (include/nlt_hip.h: nlt_conv_tile_backward_data); split as conv_backward_data's
    (transposed k2s2 mode only)."""
    sc, sy, sd, sa, sp = split if split is not None else (0, None, None, 0.0, False)
    _check(lib().nlt_conv_tile_backward_data(adj_mode, _ptr(dpre), ldp, cpre, n, h, w, _ptr(packed), cout, tn, _ptr(out), ldo,
                                             _ptr(mask_src), ldm, float(mask_alpha), 1 if accumulate else 0,
                                             sc, _ptr(sy), _ptr(sd), float(sa), 1 if sp else 0, _stream()),
           'nlt_conv_tile_backward_data')


def pack_conv_tile3_weights(mode, w_keras, cin, cout, tn):
    n = lib().nlt_conv_tile3_packed_elems(mode, cin, cout, tn)
    if n <= 0:
        raise NLTError("nlt_conv_tile3_packed_elems: unsupported (mode %d, cin %d, cout %d, tn %d)" % (mode, cin, cout, tn))
    out = torch.empty(n, device=w_keras.device, dtype=torch.int16)
    _check(lib().nlt_pack_conv_tile3_weights(mode, _ptr(_dense(w_keras, 'w_keras')), cin, cout, tn, out.data_ptr(), _stream()),
           'nlt_pack_conv_tile3_weights')
    return out


def conv_tile3_forward(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, tn, out, ldo, mean_out, ldm,
                       act=True, alpha=0.3, nprod=6):
    """conv_tile_forward with fp32 operands split into three bf16 terms (precision = f32x3; nprod = 6 or 9 term products)."""
    _check(lib().nlt_conv_tile3_forward(mode, nprod, _ptr(src), ld, cin, frames, kobs, h, w, packed.data_ptr(), _ptr(bias), cout, tn,
                                        _ptr(out), ldo, _ptr(mean_out), ldm, 1 if act else 0, float(alpha), _stream()),
           'nlt_conv_tile3_forward')


def conv_bf16_pack(mode, w_keras, c0, c1, cout):
    n = lib().nlt_conv_bf16_packed_elems(mode, c0, c1, cout)
    if n <= 0:
        raise NLTError("conv_bf16: unsupported (mode %d, c0 %d, c1 %d, cout %d)" % (mode, c0, c1, cout))
    out = torch.empty(n, device=w_keras.device, dtype=torch.bfloat16)
    _check(lib().nlt_conv_bf16_pack(mode, _ptr(_dense(w_keras, 'w_keras')), c0, c1, cout, out.data_ptr(), _stream()), 'nlt_conv_bf16_pack')
    return out


def _any_ptr(t, what):
    """(device pointer, is_fp32) of a float32 or bfloat16 CUDA tensor (or an already offset view of one)."""
    if t is None:
        return None, 0
    if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16)):
        raise NLTError("%s: expected a float32 / bfloat16 CUDA tensor, got %s on %s" % (what, t.dtype, t.device))
    return t.data_ptr(), 1 if t.dtype == torch.float32 else 0


def conv_bf16_forward(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, w_packed, bias, cout, out, ldo, act=True, alpha=0.3, tile_hint=0):
    """src* / out: float32 or bfloat16 tensors (views into wider NHWC tensors allowed: pass the per-texel stride ld*)."""
    p0, f0 = _any_ptr(src0, 'src0')
    p1, f1 = _any_ptr(src1, 'src1')
    po, fo = _any_ptr(out, 'out')
    _check(lib().nlt_conv_bf16_forward(mode, tile_hint, p0, ld0, c0, f0, p1, ld1, c1, f1, n, h, w,
                                       _tptr(w_packed, torch.bfloat16, 'w_packed'), _ptr(bias), cout, po, ldo, fo,
                                       1 if act else 0, float(alpha), _stream()), 'nlt_conv_bf16_forward')


def obs_mean_bf16(obs, n, k, hw, c, out, ldo):
    if obs.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
        raise NLTError("obs_mean_bf16: bfloat16 tensors expected")
    _check(lib().nlt_obs_mean_bf16(obs.data_ptr(), n, k, hw, c, out.data_ptr(), ldo, _stream()), 'nlt_obs_mean_bf16')


# ---------------------------------------------------------------- bf16 channel mix
def chmix_bf16_pack(w_keras):
    """Keras (1,1,cin,cout) fp32 kernel -> bf16 MFMA fragments."""
    cin, cout = w_keras.shape[2], w_keras.shape[3]
    n = lib().nlt_chmix_bf16_packed_elems(cin, cout)
    if n <= 0:
        raise NLTError("chmix_bf16: unsupported channel counts %d -> %d" % (cin, cout))
    out = torch.empty(n, device=w_keras.device, dtype=torch.bfloat16)
    _check(lib().nlt_chmix_bf16_pack(_ptr(_dense(w_keras, 'w_keras')), cin, cout, out.data_ptr(), _stream()), 'nlt_chmix_bf16_pack')
    return out


def chmix_bf16_forward(x, packed, bias, cout, act=True, alpha=0.3):
    """x [..., cin] bf16 (dense NHWC) -> [..., cout] bf16."""
    cin = x.shape[-1]
    out = torch.empty(tuple(x.shape[:-1]) + (cout,), device=x.device, dtype=torch.bfloat16)
    _check(lib().nlt_chmix_bf16_forward(_tptr(x, torch.bfloat16, 'x'), x.numel() // cin, cin,
                                        _tptr(packed, torch.bfloat16, 'packed'), _ptr(bias), cout, 1 if act else 0, float(alpha),
                                        out.data_ptr(), _stream()), 'nlt_chmix_bf16_forward')
    return out


# ---------------------------------------------------------------- fused inference ends
def front_pack_weights(wq0, bq0, wo0, bo0, wqa, bqa, wqb, bqb, woa, boa, wob, bob, wh, bh, out=None):
    """out: a blob from an earlier call to refill in place (keeps its address: launch tapes, hipGraphs)."""
    if out is None:
        out = torch.empty(lib().nlt_front_packed_floats(), device=wq0.device, dtype=torch.float32)
    args = [_ptr(_dense(t, 'weight')) for t in (wq0, bq0, wo0, bo0, wqa, bqa, wqb, bqb, woa, boa, wob, bob, wh, bh)]
    _check(lib().nlt_front_pack_weights(*args, _ptr(out), _stream()), 'nlt_front_pack_weights')
    return out


def front_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, packed, add_base, alpha, fm1, obs1, skip3):
    for t, nm in ((base, 'base'), (cvis, 'cvis'), (lvis, 'lvis'), (nn_rgb, 'nn_rgb'), (nn_base, 'nn_base')):
        _dense(t, nm)
    _check(lib().nlt_front_forward(_ptr(base), _ptr(cvis), _ptr(lvis), _ptr(nn_rgb), _ptr(nn_base), n, k, h, w,
                                   _ptr(packed), 1 if add_base else 0, float(alpha), _ptr(fm1), _ptr(obs1), _ptr(skip3),
                                   _stream()), 'nlt_front_forward')


def front_pack_l2_weights(wq, bq, wo, bo, out=None):
    if out is None:
        out = torch.empty(lib().nlt_front_l2_packed_floats(), device=wq.device, dtype=torch.float32)
    _check(lib().nlt_front_pack_l2_weights(_ptr(_dense(wq, 'wq')), _ptr(bq), _ptr(_dense(wo, 'wo')), _ptr(bo), _ptr(out),
                                           _stream()), 'nlt_front_pack_l2_weights')
    return out


def front2_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, packed, packed_l2, add_base, alpha, fm1, skip3, qtmp2, otmp2):
    for t, nm in ((base, 'base'), (cvis, 'cvis'), (lvis, 'lvis'), (nn_rgb, 'nn_rgb'), (nn_base, 'nn_base')):
        _dense(t, nm)
    _check(lib().nlt_front2_forward(_ptr(base), _ptr(cvis), _ptr(lvis), _ptr(nn_rgb), _ptr(nn_base), n, k, h, w,
                                    _ptr(packed), _ptr(packed_l2), 1 if add_base else 0, float(alpha), _ptr(fm1), _ptr(skip3),
                                    _ptr(qtmp2), _ptr(otmp2), _stream()), 'nlt_front2_forward')


def front4_supported(*tensors):
    """The LDS-staged front kernel loads 16-byte row pieces: every input buffer must start on a 16-byte boundary."""
    return all(t.data_ptr() % 16 == 0 for t in tensors)


def front4_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, packed, packed_l2, add_base, alpha, fm1, skip3, qtmp2, otmp2,
                   waves_per_simd=0):
    for t, nm in ((base, 'base'), (cvis, 'cvis'), (lvis, 'lvis'), (nn_rgb, 'nn_rgb'), (nn_base, 'nn_base')):
        _dense(t, nm)
    _check(lib().nlt_front4_forward(_ptr(base), _ptr(cvis), _ptr(lvis), _ptr(nn_rgb), _ptr(nn_base), n, k, h, w,
                                    _ptr(packed), _ptr(packed_l2), 1 if add_base else 0, float(alpha), _ptr(fm1), _ptr(skip3),
                                    _ptr(qtmp2), _ptr(otmp2), int(waves_per_simd), _stream()), 'nlt_front4_forward')


def front4_forward_u8(diffuse_store, rgb_store, cvis_store, lvis_store, ids, nn_ids, n, k, h, w, packed, packed_l2, add_base,
                      alpha, fm1, skip3, qtmp2, otmp2, waves_per_simd=0):
    u8 = torch.uint8
    _check(lib().nlt_front4_forward_u8(_tptr(diffuse_store, u8, 'diffuse_store'), _tptr(rgb_store, u8, 'rgb_store'),
                                       _tptr(cvis_store, u8, 'cvis_store'), _tptr(lvis_store, u8, 'lvis_store'),
                                       _tptr(ids, torch.int32, 'ids'), _tptr(nn_ids, torch.int32, 'nn_ids'), n, k, h, w,
                                       _ptr(packed), _ptr(packed_l2), 1 if add_base else 0, float(alpha), _ptr(fm1),
                                       _ptr(skip3), _ptr(qtmp2), _ptr(otmp2), int(waves_per_simd), _stream()),
           'nlt_front4_forward_u8')


def front4_forward_train(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, packed, packed_l2, add_base, alpha, fm1, skip3, qtmp2, otmp2,
                         obs1, qtmp1, otmp1):
    """front4_forward that also keeps the level-1 maps the backward pass reads (train mode)."""
    _check(lib().nlt_front4_forward_train(_ptr(base), _ptr(cvis), _ptr(lvis), _ptr(nn_rgb), _ptr(nn_base), n, k, h, w, _ptr(packed),
                                          _ptr(packed_l2), 1 if add_base else 0, float(alpha), _ptr(fm1), _ptr(skip3), _ptr(qtmp2),
                                          _ptr(otmp2), _ptr(obs1), _ptr(qtmp1), _ptr(otmp1), _stream()), 'nlt_front4_forward_train')


def front_ovr_forward(base, cvis, lvis, n, h, w, packed, packed_l2, p1, s0, p2, add_base, alpha, q1, ldq, skip3, qtmp2):
    """The query-only front launch of the reference's inference mode (include/nlt_hip.h: nlt_front_ovr_forward)."""
    for t, nm in ((base, 'base'), (cvis, 'cvis'), (lvis, 'lvis'), (p1, 'p1'), (s0, 's0'), (p2, 'p2')):
        _dense(t, nm)
    _check(lib().nlt_front_ovr_forward(_ptr(base), _ptr(cvis), _ptr(lvis), n, h, w, _ptr(packed), _ptr(packed_l2), _ptr(p1),
                                       _ptr(s0), _ptr(p2), 1 if add_base else 0, float(alpha), _ptr(q1), ldq, _ptr(skip3),
                                       _ptr(qtmp2), _stream()), 'nlt_front_ovr_forward')


def front_ovr_forward_u8(diffuse_store, cvis_store, lvis_store, ids, n, h, w, packed, packed_l2, p1, s0, p2, add_base, alpha, q1, ldq,
                         skip3, qtmp2):
    """front_ovr_forward on a store-resident batch: frame ids of the uint8 capture store (nlt_front_ovr_forward_u8)."""
    u8 = torch.uint8
    for t, nm in ((p1, 'p1'), (s0, 's0'), (p2, 'p2')):
        _dense(t, nm)
    _check(lib().nlt_front_ovr_forward_u8(_tptr(diffuse_store, u8, 'diffuse_store'), _tptr(cvis_store, u8, 'cvis_store'),
                                          _tptr(lvis_store, u8, 'lvis_store'), _tptr(ids, torch.int32, 'ids'), n, h, w,
                                          _ptr(packed), _ptr(packed_l2), _ptr(p1), _ptr(s0), _ptr(p2), 1 if add_base else 0,
                                          float(alpha), _ptr(q1), ldq, _ptr(skip3), _ptr(qtmp2), _stream()),
           'nlt_front_ovr_forward_u8')


def dec_block_forward_map(x, skip, lds, n, h, w, w_s2q, w_s1, b_s1, c, alpha, bias_map, out):
    """One expanding block of the inference mode: [x 2c | query half 4c of `skip`] + bias map (include/nlt_hip.h)."""
    _check(lib().nlt_dec_block_forward_map(_ptr(_dense(x, 'x')), _ptr(skip), lds, n, h, w, _ptr(_dense(w_s2q, 'w_s2q')),
                                           _ptr(_dense(w_s1, 'w_s1')), _ptr(b_s1), c, float(alpha), _ptr(_dense(bias_map, 'bias_map')),
                                           _ptr(_dense(out, 'out')), _stream()), 'nlt_dec_block_forward_map')


def back_forward_map(x, q1, ldq, skip3, n, h2, w2, w_s2q, w_s1, b_s1, w_head, alpha, bias_map, pred):
    """Last expanding block + head of the inference mode: [x 8 | 16 query channels of the level-1 map] + bias map."""
    _check(lib().nlt_back_forward_map(_ptr(_dense(x, 'x')), _ptr(q1), ldq, _ptr(_dense(skip3, 'skip3')), n, h2, w2,
                                      _ptr(_dense(w_s2q, 'w_s2q')), _ptr(_dense(w_s1, 'w_s1')), _ptr(b_s1), _ptr(_dense(w_head, 'w_head')),
                                      float(alpha), _ptr(_dense(bias_map, 'bias_map')), _ptr(_dense(pred, 'pred')), _stream()),
           'nlt_back_forward_map')


def dec_block_forward(x, cx, skip, cs, n, h, w, w_s2, b_s2, w_s1, b_s1, c, alpha, out):
    """One expanding block (deconv k2s2 + LeakyReLU + deconv k2s1 + LeakyReLU), c = 8 or 16, intermediate in LDS."""
    _check(lib().nlt_dec_block_forward(_ptr(_dense(x, 'x')), cx, _ptr(_dense(skip, 'skip')), cs, n, h, w, _ptr(w_s2), _ptr(b_s2),
                                       _ptr(w_s1), _ptr(b_s1), c, float(alpha), _ptr(out), _stream()), 'nlt_dec_block_forward')


def back_forward(x, fm1, skip3, n, h2, w2, w_s2, b_s2, w_s1, b_s1, w_head, alpha, pred):
    _check(lib().nlt_back_forward(_ptr(_dense(x, 'x')), _ptr(_dense(fm1, 'fm1')), _ptr(_dense(skip3, 'skip3')), n, h2, w2,
                                  _ptr(w_s2), _ptr(b_s2), _ptr(w_s1), _ptr(b_s1), _ptr(w_head), float(alpha), _ptr(pred),
                                  _stream()), 'nlt_back_forward')


def front_forward_train(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, packed, add_base, alpha, fm1, obs1, skip3, qtmp1, otmp1):
    for t, nm in ((base, 'base'), (cvis, 'cvis'), (lvis, 'lvis'), (nn_rgb, 'nn_rgb'), (nn_base, 'nn_base')):
        _dense(t, nm)
    _check(lib().nlt_front_forward_train(_ptr(base), _ptr(cvis), _ptr(lvis), _ptr(nn_rgb), _ptr(nn_base), n, k, h, w,
                                         _ptr(packed), 1 if add_base else 0, float(alpha), _ptr(fm1), _ptr(obs1),
                                         _ptr(skip3), _ptr(_dense(qtmp1, 'qtmp1')), _ptr(_dense(otmp1, 'otmp1')),
                                         _stream()), 'nlt_front_forward_train')


def back_forward_train(x, fm1, skip3, n, h2, w2, w_s2, b_s2, w_s1, b_s1, w_head, alpha, pred, u, v):
    _check(lib().nlt_back_forward_train(_ptr(_dense(x, 'x')), _ptr(_dense(fm1, 'fm1')), _ptr(_dense(skip3, 'skip3')),
                                        n, h2, w2, _ptr(w_s2), _ptr(b_s2), _ptr(w_s1), _ptr(b_s1), _ptr(w_head),
                                        float(alpha), _ptr(pred), _ptr(_dense(u, 'u')), _ptr(_dense(v, 'v')), _stream()),
           'nlt_back_forward_train')


def front_backward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, dy1q, dy1o, dpred, weights, grads):
    """weights = (wq0, bq0, wo0, bo0, wqa, woa, wh); grads = (dwq0, dbq0, dwo0, dbo0, dwqa, dbqa, dwoa, dboa, dwh),
    accumulated in place (views of the flat gradient bucket)."""
    need = lib().nlt_front_backward_workspace_floats(n, h, w)
    if need < 0:
        raise NLTError("nlt_front_backward: unsupported shape %dx%d" % (h, w))
    ws = _workspace('front_bwd', base.device, need)
    for t, nm in ((base, 'base'), (cvis, 'cvis'), (lvis, 'lvis'), (nn_rgb, 'nn_rgb'), (nn_base, 'nn_base'),
                  (dy1q, 'dy1q'), (dy1o, 'dy1o'), (dpred, 'dpred')):
        _dense(t, nm)
    args = [_ptr(t) for t in (base, cvis, lvis, nn_rgb, nn_base)] + [n, k, h, w, _ptr(dy1q), _ptr(dy1o), _ptr(dpred)]
    args += [_ptr(_dense(t, 'weight')) for t in weights] + [_ptr(_dense(t, 'grad')) for t in grads]
    _check(lib().nlt_front_backward(*args, _ptr(ws), _stream()), 'nlt_front_backward')


def back_backward(x, fm1, u, v, dpred, n, h2, w2, w_s2, w_s1, w_head, alpha, dx, dfm1, dw_s2, db_s2, dw_s1, db_s1, dw_head, db_head):
    """Gradients of the last expanding block + head; dx / dfm1 written, weight gradients accumulated in place."""
    need = lib().nlt_back_backward_workspace_floats(n, h2, w2)
    if need <= 0:
        raise NLTError("nlt_back_backward_workspace_floats(%d,%d,%d) failed" % (n, h2, w2))
    ws = _workspace('back_bwd', x.device, need)
    ins = [_ptr(_dense(t, nm)) for t, nm in ((x, 'x'), (fm1, 'fm1'), (u, 'u'), (v, 'v'), (dpred, 'dpred'))]
    wts = [_ptr(_dense(t, 'weight')) for t in (w_s2, w_s1, w_head)]
    outs = [_ptr(_dense(t, 'grad')) for t in (dx, dfm1, dw_s2, db_s2, dw_s1, db_s1, dw_head, db_head)]
    _check(lib().nlt_back_backward(*ins, n, h2, w2, *wts, float(alpha), *outs, _ptr(ws), _stream()), 'nlt_back_backward')


# ---------------------------------------------------------------- texel-buffer assembly
_MAP_DTYPES = {torch.float64: MAP_F64, torch.float32: MAP_F32, torch.float16: MAP_F16}


def cosine_map(locs, normals, valid, occluded, src_loc, want_float=True, want_u8=True):
    """locs/normals [...,3] float64, valid/occluded [...] uint8 -> (cos float64, quantised uint8)."""
    shape = valid.shape
    pixels = valid.numel()
    cos = torch.empty(shape, device=valid.device, dtype=torch.float64) if want_float else None
    q = torch.empty(shape, device=valid.device, dtype=torch.uint8) if want_u8 else None
    sx, sy, sz = (float(x) for x in src_loc)
    _check(lib().nlt_cosine_map(_tptr(locs, torch.float64, 'locs'), _tptr(normals, torch.float64, 'normals'),
                                _tptr(valid, torch.uint8, 'valid'), _tptr(occluded, torch.uint8, 'occluded'),
                                sx, sy, sz, pixels, _tptr(cos, torch.float64, 'cos'), _tptr(q, torch.uint8, 'q'),
                                _stream()), 'nlt_cosine_map')
    return cos, q


def albedo(rgb_frames):
    """rgb_frames [F,H,W,3] uint8 -> albedo [H,W,3] float64."""
    f = rgb_frames.shape[0]
    elems = rgb_frames[0].numel()
    out = torch.empty(rgb_frames.shape[1:], device=rgb_frames.device, dtype=torch.float64)
    ws = torch.empty(1, device=rgb_frames.device, dtype=torch.int64)
    _check(lib().nlt_albedo(_tptr(rgb_frames, torch.uint8, 'rgb_frames'), f, elems, out.data_ptr(), ws.data_ptr(),
                            _stream()), 'nlt_albedo')
    return out


def diffuse_base(albedo_, lvis):
    """albedo [H,W,3] float64, lvis [F,H,W] uint8 -> diffuse [F,H,W,3] uint8."""
    f = lvis.shape[0]
    texels = lvis[0].numel()
    out = torch.empty(tuple(lvis.shape) + (3,), device=lvis.device, dtype=torch.uint8)
    _check(lib().nlt_diffuse_base(_tptr(albedo_, torch.float64, 'albedo'), _tptr(lvis, torch.uint8, 'lvis'), f, texels,
                                  out.data_ptr(), _stream()), 'nlt_diffuse_base')
    return out


def remap_bilinear(src, mapping, force_kbg=True):
    """src [h,w] or [h,w,c] uint8 / float32; mapping [oh,ow,>=2] float64/32/16 in [0,1] -> [oh,ow(,c)]."""
    if mapping.dtype not in _MAP_DTYPES:
        raise NLTError("mapping dtype %s" % mapping.dtype)
    squeeze = src.dim() == 2
    h, w = src.shape[:2]
    c = 1 if squeeze else src.shape[2]
    oh, ow, ldm = mapping.shape
    out = torch.empty((oh, ow) if squeeze else (oh, ow, c), device=src.device, dtype=src.dtype)
    if src.dtype == torch.uint8:
        fn, what = lib().nlt_remap_bilinear_u8, 'nlt_remap_bilinear_u8'
    elif src.dtype == torch.float32:
        fn, what = lib().nlt_remap_bilinear_f32, 'nlt_remap_bilinear_f32'
    else:
        raise NLTError("remap source dtype %s" % src.dtype)
    _check(fn(_tptr(src, src.dtype, 'src'), h, w, c, _tptr(mapping, mapping.dtype, 'mapping'), _MAP_DTYPES[mapping.dtype],
              ldm, oh, ow, 1 if force_kbg else 0, out.data_ptr(), _stream()), what)
    return out


def uv_index_map(uvs, values, h, w, max_l1=4, fill=0.0, want_index=False):
    """uvs [P,2], values [P,M] float64 -> grid [h,w,M] float64 (+ int32 sample-index map)."""
    p, m = values.shape
    nbytes = lib().nlt_uv_index_map_workspace_bytes(h, w, p)
    if nbytes <= 0:
        raise NLTError("nlt_uv_index_map_workspace_bytes(%d,%d,%d) failed" % (h, w, p))
    ws = torch.empty((nbytes + 3) // 4, device=uvs.device, dtype=torch.int32)
    out = torch.empty((h, w, m), device=uvs.device, dtype=torch.float64)
    idx = torch.empty((h, w), device=uvs.device, dtype=torch.int32) if want_index else None
    _check(lib().nlt_uv_index_map(_tptr(uvs, torch.float64, 'uvs'), _tptr(values, torch.float64, 'values'), p, m, h, w,
                                  int(max_l1), float(fill), ws.data_ptr(), out.data_ptr(),
                                  _tptr(idx, torch.int32, 'index_out'), _stream()), 'nlt_uv_index_map')
    return (out, idx) if want_index else out


def knn_indices(ref_pos, cand_pos, k=1):
    """ref_pos [P,3], cand_pos [Q,3] float64 -> int32 [P,k]."""
    p, q = ref_pos.shape[0], cand_pos.shape[0]
    out = torch.empty((p, k), device=ref_pos.device, dtype=torch.int32)
    _check(lib().nlt_knn_indices(_tptr(ref_pos, torch.float64, 'ref_pos'), p, _tptr(cand_pos, torch.float64, 'cand_pos'),
                                 q, k, out.data_ptr(), _stream()), 'nlt_knn_indices')
    return out


def psnr_sums(im1, im2, mask=None):
    """im1 / im2 [H,W] or [H,W,C] float32 (C = 1 or 3), mask [H,W] uint8 or None -> (sum of squared luma differences,
    number of pixels counted) as a float64 CUDA tensor [2]."""
    _same_shape(im1, im2, 'psnr_sums')
    c = 1 if im1.dim() == 2 else im1.shape[2]
    pixels = im1.numel() // c
    if mask is not None and mask.numel() != pixels:
        raise NLTError("psnr_sums: mask has %d pixels, the images %d" % (mask.numel(), pixels))
    ws = torch.empty(512, device=im1.device, dtype=torch.float64)
    out = torch.empty(2, device=im1.device, dtype=torch.float64)
    _check(lib().nlt_psnr_sums(_ptr(_dense(im1, 'im1')), _ptr(_dense(im2, 'im2')), _tptr(mask, torch.uint8, 'mask'), pixels, c,
                               ws.data_ptr(), out.data_ptr(), _stream()), 'nlt_psnr_sums')
    return out


def gather_frames_u8(store, ids, out=None):
    """store [F,...] uint8, ids [n] int32 (-1 -> zeros) -> float32 [n,...] = float32(float64(u8) / 255).
    out: a persistent destination of that shape (a loader's staging ring) instead of a fresh tensor."""
    n = ids.numel()
    shape = (n,) + tuple(store.shape[1:])
    if out is None:
        out = torch.empty(shape, device=store.device, dtype=torch.float32)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise NLTError("gather_frames_u8: out must be a contiguous float32 tensor of shape %s" % (shape,))
    _check(lib().nlt_gather_frames_u8(_tptr(store, torch.uint8, 'store'), _tptr(ids, torch.int32, 'ids'), n,
                                      store[0].numel(), _ptr(out), _stream()), 'nlt_gather_frames_u8')
    return out


def assemble_batch(diffuse_store, rgb_store, cvis_store, lvis_store, ids, nn_ids, test_mode=False, out=None):
    """uint8 stores [F,H,W,3] / [F,H,W]; ids [N] int32; nn_ids [N,k] int32 -> dict of float32 buffers.
    out: the dict of an earlier call with the same shapes, refilled in place (a loader's staging ring)."""
    n = ids.numel()
    k = 0 if nn_ids is None else nn_ids.shape[1]
    _, h, w = cvis_store.shape
    dev = cvis_store.device
    E = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
    if out is None:
        out = {'base': E(n, h, w, 3), 'cvis': E(n, h, w, 1), 'lvis': E(n, h, w, 1), 'rgb': E(n, h, w, 3),
               'nn_base': E(n, k, h, w, 3) if k else None, 'nn_rgb': E(n, k, h, w, 3) if k else None}
    elif tuple(out['base'].shape) != (n, h, w, 3) or (k and tuple(out['nn_rgb'].shape) != (n, k, h, w, 3)):
        raise NLTError("assemble_batch: `out` was made for another batch shape")
    u8 = torch.uint8
    _check(lib().nlt_assemble_batch(_tptr(diffuse_store, u8, 'diffuse_store'), _tptr(rgb_store, u8, 'rgb_store'),
                                    _tptr(cvis_store, u8, 'cvis_store'), _tptr(lvis_store, u8, 'lvis_store'),
                                    _tptr(ids, torch.int32, 'ids'), _tptr(nn_ids, torch.int32, 'nn_ids'), n, k, h * w,
                                    1 if test_mode else 0, *[_ptr(out[x]) for x in ('base', 'cvis', 'lvis', 'rgb', 'nn_base', 'nn_rgb')],
                                    _stream()), 'nlt_assemble_batch')
    return out
