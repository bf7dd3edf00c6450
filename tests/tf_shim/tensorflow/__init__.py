"""TEST-ONLY stand-in for the slice of TensorFlow 2.2 that the reference's model code touches (see ../README.md).
Tensors wrap torch CPU tensors; every numeric primitive delegates to oracle/tf_ops.py."""
import types

import numpy as np
import torch

from oracle import tf_ops as T

float32, float64, int32 = torch.float32, torch.float64, torch.int32
newaxis = None


class TensorShape(tuple):
    def as_list(self):
        return list(self)


class Tensor:
    """Eager tensor: immutable value semantics are all the reference relies on."""

    def __init__(self, t):
        self.t = t.t if isinstance(t, Tensor) else torch.as_tensor(t)

    shape = property(lambda self: TensorShape(self.t.shape))
    dtype = property(lambda self: self.t.dtype)

    def get_shape(self):
        return self.shape

    def numpy(self):
        return self.t.detach().numpy()

    def __len__(self):
        return self.t.shape[0]

    def __getitem__(self, idx):
        return Tensor(self.t[idx])

    def _bin(self, other, fn, swap=False):
        o = other.t if isinstance(other, Tensor) else other
        return Tensor(fn(o, self.t) if swap else fn(self.t, o))

    __add__ = lambda s, o: s._bin(o, torch.add)
    __radd__ = lambda s, o: s._bin(o, torch.add, True)
    __sub__ = lambda s, o: s._bin(o, torch.sub)
    __rsub__ = lambda s, o: s._bin(o, lambda a, b: a - b, True)
    __mul__ = lambda s, o: s._bin(o, torch.mul)
    __rmul__ = lambda s, o: s._bin(o, torch.mul, True)
    __truediv__ = lambda s, o: s._bin(o, torch.div)
    __neg__ = lambda s: Tensor(-s.t)


def _t(x, dtype=None):
    if isinstance(x, Tensor):
        return x.t
    t = torch.as_tensor(np.asarray(x)) if not torch.is_tensor(x) else x
    return t.to(dtype) if dtype is not None else t


def convert_to_tensor(x, dtype=None):
    return Tensor(_t(x, dtype))


def cast(x, dtype):
    return Tensor(_t(x).to(dtype))


def concat(values, axis):
    return Tensor(torch.cat([_t(v) for v in values], axis))


def stack(values, axis=0):
    return Tensor(torch.stack([_t(v) for v in values], axis))


def expand_dims(x, axis):
    return Tensor(_t(x).unsqueeze(axis))


def reshape(x, shape):
    return Tensor(_t(x).reshape(tuple(shape)))


def tile(x, multiples):
    return Tensor(_t(x).repeat(*multiples))


def ones(shape, dtype=float32):
    return Tensor(torch.ones(tuple(shape), dtype=dtype))


def zeros(shape, dtype=float32):
    return Tensor(torch.zeros(tuple(shape), dtype=dtype))


def multiply(a, b):
    return Tensor(_t(a) * _t(b))


def reduce_mean(x, axis=None, keepdims=False):
    t = _t(x)
    if axis is None:
        return Tensor(t.mean())
    if isinstance(axis, (tuple, list)) and len(axis) == 0:
        return Tensor(t)
    return Tensor(t.mean(dim=axis, keepdim=keepdims))


def clip_by_value(x, lo, hi):
    return Tensor(_t(x).clamp(lo, hi))


class _Layer:
    """tf.keras.layers.Layer: builds on first call, like Keras."""
    built = False

    def build(self, input_shape):
        self.built = True

    def __call__(self, x, **unused):
        if not self.built:
            self.build(tuple(x.shape))
            self.built = True
        return Tensor(self.call(_t(x)))

    @property
    def trainable_variables(self):
        return [v for v in (getattr(self, 'kernel', None), getattr(self, 'bias', None)) if v is not None]


class _ConvBase(_Layer):
    transpose = False

    def __init__(self, filters, kernel_size, strides=1, padding='valid'):
        assert padding == 'same', "the reference only builds padding='same' convs (nlt/networks/elements.py:26-39)"
        self.filters, self.kernel_size, self.strides = filters, kernel_size, strides
        self.kernel = self.bias = None

    def build(self, input_shape):
        cin, k = input_shape[-1], self.kernel_size
        shape = (k, k, self.filters, cin) if self.transpose else (k, k, cin, self.filters)
        # Keras defaults: glorot_uniform kernel, zeros bias (the generator script overwrites both)
        self.kernel = torch.from_numpy(T.glorot_uniform(np.random.default_rng(0), shape).astype(np.float32))
        self.bias = torch.zeros(self.filters)
        self.built = True

    def set_weights(self, weights):
        k, b = weights
        self.kernel, self.bias = torch.as_tensor(np.asarray(k)), torch.as_tensor(np.asarray(b))
        self.built = True


class Conv2D(_ConvBase):
    def call(self, x):
        return T.conv2d_same(x, self.kernel, self.bias, self.strides)


class Conv2DTranspose(_ConvBase):
    transpose = True

    def call(self, x):
        return T.conv2d_transpose_same(x, self.kernel, self.bias, self.strides)


class LeakyReLU(_Layer):
    def __init__(self, alpha=0.3):
        self.alpha = alpha

    def call(self, x):
        return T.leaky_relu(x, self.alpha)


class ReLU(_Layer):
    def __init__(self, negative_slope=0):
        self.alpha = negative_slope

    def call(self, x):
        return T.leaky_relu(x, self.alpha)


class ELU(_Layer):
    def __init__(self, alpha=1.0):
        self.alpha = alpha

    def call(self, x):
        return torch.where(x > 0, x, self.alpha * (torch.exp(x) - 1))


class Lambda(_Layer):
    def __init__(self, fn):
        self.fn = fn

    def call(self, x):
        return _t(self.fn(Tensor(x)))


class Sequential(_Layer):
    def __init__(self, layers=None):
        self.layers = list(layers or [])

    def build(self, input_shape):
        x = Tensor(torch.zeros((1,) + tuple(input_shape[1:])))
        for l in self.layers:
            x = l(x)
        self.built = True

    @property
    def built(self):
        return all(l.built for l in self.layers)

    @built.setter
    def built(self, v):
        pass

    def __call__(self, x, **unused):
        for l in self.layers:
            x = l(x)
        return x

    @property
    def trainable_variables(self):
        return [v for l in self.layers for v in l.trainable_variables]


class Model:
    """tf.keras.Model, "used only for the parent's trackability" (nlt/models/base.py:27)."""

    def __init__(self, *a, **kw):
        pass

    def __call__(self, *a, **kw):
        return self.call(*a, **kw)

    @property
    def trainable_variables(self):
        out = []
        for name in sorted(vars(self)):
            v = getattr(self, name)
            if isinstance(v, _Layer):
                out += v.trainable_variables
        return out


class MeanSquaredError:
    """tf.keras.losses.MeanSquaredError(reduction='none'): mean over the LAST axis only."""

    def __init__(self, reduction='none'):
        assert reduction == 'none'

    def __call__(self, y_true, y_pred, sample_weight=None):
        loss = ((_t(y_pred) - _t(y_true)) ** 2).mean(-1)
        if sample_weight is not None:
            loss = loss * _t(sample_weight)
        return Tensor(loss)


class MeanAbsoluteError(MeanSquaredError):
    def __call__(self, y_true, y_pred, sample_weight=None):
        loss = (_t(y_pred) - _t(y_true)).abs().mean(-1)
        return Tensor(loss if sample_weight is None else loss * _t(sample_weight))


def _resize(x, size):
    """tf.image.resize, TF2 defaults (bilinear, half-pixel centres, no antialias); returns float32 like TF."""
    return Tensor(T.resize_bilinear(_t(x), int(size[0]), int(size[1])))


def _no_op(*a, **kw):
    return None


compat = types.SimpleNamespace(v1=types.SimpleNamespace(enable_eager_execution=_no_op))
keras = types.SimpleNamespace(
    Model=Model, Sequential=Sequential,
    layers=types.SimpleNamespace(Conv2D=Conv2D, Conv2DTranspose=Conv2DTranspose, LeakyReLU=LeakyReLU, ReLU=ReLU, ELU=ELU,
                                 Lambda=Lambda),
    losses=types.SimpleNamespace(MeanSquaredError=MeanSquaredError, MeanAbsoluteError=MeanAbsoluteError))
image = types.SimpleNamespace(resize=_resize)
