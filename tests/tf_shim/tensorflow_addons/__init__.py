"""TEST-ONLY stand-in for tensorflow_addons 0.10 (see ../README.md): image.resampler -> oracle/tf_ops.resampler."""
import types

import tensorflow as tf
from oracle import tf_ops as T


def _resampler(data, warp):
    return tf.Tensor(T.resampler(tf._t(data), tf._t(warp)))


image = types.SimpleNamespace(resampler=_resampler)
