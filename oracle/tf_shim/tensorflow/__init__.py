"""A `tensorflow`-named NumPy shim -- TEST INFRASTRUCTURE, this container only.

Purpose: let the reference's own ``modules/*.py`` and ``models/models.py`` (which ``import tensorflow as tf``)
be imported from /root/reference and executed eagerly, so that the COMPOSITION of the oracle
(oracle/vaenar_numpy.py: call order, concat order, masks, head swaps, reshapes ...) can be cross-checked
against the reference's own Python.  It implements only the ~75 tf symbols the text->mel path touches, with
the documented TF 2.2 semantics, in float64 ("float32" maps to float64 so that the
comparison isolates composition from round-off).  It does NOT pin TensorFlow's kernel numerics: parity with
the real reference remains unpinned (see oracle/vaenar_numpy.py header).  Never shipped to the GPU box's
product path, never imported by vaenar_tts_amd.
"""
import sys
import types

import numpy as np

_F = np.float64
float32 = "float32"
float64 = "float64"
int32 = "int32"
int64 = "int64"
bool = "bool"  # noqa: A001  (tf.bool)


def _dt(d):
    if d is None:
        return None
    if d in ("float32", "float64", float32) or d is np.float32 or d is np.float64 or d is float:
        return _F
    if d in ("int32", "int64") or d is int:
        return np.int64
    if d in ("bool",):
        return np.bool_
    return np.dtype(d).type


class Tensor(np.ndarray):
    def __new__(cls, a):
        return np.asarray(a).view(cls)

    def set_shape(self, shape):
        return None

    def numpy(self):
        return np.asarray(self)

    def assign(self, v):
        self[...] = np.asarray(v)
        return self


def _t(a):
    return Tensor(np.asarray(a))


class TensorShape(list):
    pass


class Variable(Tensor):
    def __new__(cls, initial_value, trainable=True, dtype=None, name=None):
        a = np.array(initial_value, dtype=_F if np.asarray(initial_value).dtype.kind == "f" else None)
        obj = a.view(cls)
        obj.trainable = trainable
        return obj

    def __array_finalize__(self, obj):
        self.trainable = getattr(obj, "trainable", True)


def constant(v, dtype=None, shape=None, name=None):
    a = np.asarray(v, dtype=_dt(dtype)) if dtype is not None else np.asarray(v)
    if a.dtype.kind == "f":
        a = a.astype(_F)
    return _t(a)


def shape(x):
    return _t(np.array(np.shape(x), dtype=np.int64))


def reshape(x, s):
    return _t(np.reshape(x, [int(i) for i in np.asarray(s).reshape(-1)]))


def transpose(x, perm=None):
    return _t(np.transpose(x, perm))


def tile(x, m):
    return _t(np.tile(x, [int(i) for i in np.asarray(m).reshape(-1)]))


def expand_dims(x, axis):
    return _t(np.expand_dims(x, int(axis)))


def concat(xs, axis):
    return _t(np.concatenate([np.asarray(x) for x in xs], int(axis)))


def split(x, num_or_size_splits, axis=0):
    return [_t(p) for p in np.split(np.asarray(x), num_or_size_splits, int(axis))]


def cast(x, dtype):
    d = _dt(dtype)
    a = np.asarray(x)
    if d is np.int64 and a.dtype.kind == "f":
        a = np.trunc(a)                       # tf.cast(float -> int32) truncates toward zero
    return _t(a.astype(d))


def range(start, limit=None, delta=1, dtype=None):  # noqa: A001
    if limit is None:
        start, limit = 0, start
    return _t(np.arange(float(start) if _dt(dtype) is _F else start, limit, delta).astype(_dt(dtype) or np.int64))


def ones(s, dtype=float32):
    return _t(np.ones([int(i) for i in np.asarray(s).reshape(-1)], _dt(dtype)))


def zeros(s, dtype=float32):
    return _t(np.zeros([int(i) for i in np.asarray(s).reshape(-1)], _dt(dtype)))


def ones_like(x, dtype=None):
    return _t(np.ones(np.shape(x), _dt(dtype) or np.asarray(x).dtype))


def where(c, x=None, y=None):
    return _t(np.where(np.asarray(c), np.asarray(x), np.asarray(y)))


def sequence_mask(lengths, maxlen=None, dtype=bool, name=None):
    lengths = np.asarray(lengths)
    if maxlen is None:
        maxlen = int(lengths.max())
    m = np.arange(int(maxlen))[None, :] < lengths[..., None]
    return _t(m.astype(_dt(dtype)))


def stop_gradient(x):
    return x


def identity(x):
    return x


def matmul(a, b, transpose_a=False, transpose_b=False):
    a, b = np.asarray(a), np.asarray(b)
    if transpose_a:
        a = np.swapaxes(a, -1, -2)
    if transpose_b:
        b = np.swapaxes(b, -1, -2)
    return _t(a @ b)


def reduce_sum(x, axis=None, keepdims=False):
    return _t(np.sum(np.asarray(x), axis=tuple(axis) if isinstance(axis, (list, tuple)) else axis, keepdims=keepdims))


def reduce_mean(x, axis=None, keepdims=False):
    return _t(np.mean(np.asarray(x), axis=tuple(axis) if isinstance(axis, (list, tuple)) else axis, keepdims=keepdims))


def reduce_max(x, axis=None):
    return _t(np.max(np.asarray(x), axis=axis))


def exp(x):
    return _t(np.exp(np.asarray(x, _F)))


def sqrt(x):
    return _t(np.sqrt(np.asarray(x, _F)))


def square(x):
    return _t(np.square(np.asarray(x)))


def pow(x, y):  # noqa: A001
    return _t(np.power(np.asarray(x, _F), np.asarray(y, _F)))


def abs(x):  # noqa: A001
    return _t(np.abs(x))


def logical_and(a, b):
    return _t(np.logical_and(a, b))


def stack(xs, axis=0):
    return _t(np.stack(xs, axis))


def _sigmoid(x):
    return _t(1.0 / (1.0 + np.exp(-np.asarray(x, _F))))


def _softmax(x, axis=-1):
    x = np.asarray(x, _F)
    m = x.max(axis, keepdims=True)
    e = np.exp(x - m)
    return _t(e / e.sum(axis, keepdims=True))


math = types.SimpleNamespace(
    softmax=_softmax, sigmoid=_sigmoid, log=lambda x: _t(np.log(np.asarray(x, _F))), exp=exp, sqrt=sqrt,
    sin=lambda x: _t(np.sin(np.asarray(x, _F))), cos=lambda x: _t(np.cos(np.asarray(x, _F))),
    mod=lambda a, b: _t(np.mod(a, b)), equal=lambda a, b: _t(np.equal(a, b)), logical_and=logical_and,
    reduce_sum=reduce_sum, reduce_mean=reduce_mean, reduce_max=reduce_max,
    reduce_std=lambda x, axis=None: _t(np.std(np.asarray(x), axis=axis)),     # population std (ddof 0)
    tanh=lambda x: _t(np.tanh(np.asarray(x, _F))))
nn = types.SimpleNamespace(relu=lambda x: _t(np.maximum(np.asarray(x), 0)), tanh=math.tanh, sigmoid=_sigmoid)


def _band_part(x, lower, upper, name=None):
    a = np.asarray(x)
    r, c = a.shape[-2:]
    i, j = np.arange(r)[:, None], np.arange(c)[None, :]
    keep = np.ones((r, c), np.bool_)
    if lower >= 0:
        keep &= (i - j) <= lower
    if upper >= 0:
        keep &= (j - i) <= upper
    return _t(np.where(keep, a, np.zeros_like(a)))


linalg = types.SimpleNamespace(
    matmul=matmul, band_part=_band_part,
    slogdet=lambda a: (_t(np.linalg.slogdet(np.asarray(a, _F))[0]), _t(np.linalg.slogdet(np.asarray(a, _F))[1])),
    inv=lambda a: _t(np.linalg.inv(np.asarray(a, _F))))


class _Random:
    """tf.random.normal with an injectable queue: tests push the noise the oracle uses."""
    queue = []

    @classmethod
    def normal(cls, shape, mean=0.0, stddev=1.0, dtype=None):
        shp = tuple(int(i) for i in np.asarray(shape).reshape(-1))
        if cls.queue:
            a = np.asarray(cls.queue.pop(0), _F)
            assert a.shape == shp, (a.shape, shp)
            return _t(a * float(np.asarray(stddev)) + float(np.asarray(mean)))
        return _t(np.zeros(shp, _F) + float(np.asarray(mean)))


random = _Random
nest = types.SimpleNamespace(flatten=lambda x: list(x) if isinstance(x, (list, tuple)) else [x])
losses = types.SimpleNamespace(MeanSquaredError=object, MeanAbsoluteError=object)


# ---- tf.keras ---------------------------------------------------------------------------------------------------
_TRAINING = [None]        # Keras call-context propagation of `training`


class Layer:
    def __init__(self, name=None, **kwargs):
        self.name = name

    def __call__(self, *args, **kwargs):
        outer = _TRAINING[0]
        if kwargs.get("training", None) is not None:
            _TRAINING[0] = kwargs["training"]
        try:
            return self.call(*args, **kwargs)
        finally:
            _TRAINING[0] = outer


class Model(Layer):
    pass


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_initializer=None, name=None, **kw):
        super().__init__(name=name)
        self.units, self.use_bias = units, use_bias
        self.activation = {"relu": nn.relu, "tanh": math.tanh, None: None}.get(activation, activation) \
            if isinstance(activation, (str, type(None))) else activation
        self.kernel = self.bias = None

    def call(self, x, **kw):
        y = np.asarray(x, _F) @ np.asarray(self.kernel)
        if self.use_bias:
            y = y + np.asarray(self.bias)
        y = _t(y)
        return self.activation(y) if self.activation is not None else y


class Conv1D(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding="valid", activation=None, name=None, **kw):
        super().__init__(name=name)
        assert strides == 1 and str(padding).lower() == "same" and activation is None
        self.kernel = self.bias = None

    def call(self, x, **kw):
        x = np.asarray(x, _F)
        k, cin, cout = self.kernel.shape
        B, T, _ = x.shape
        left = (k - 1) // 2
        xp = np.zeros((B, T + k - 1, cin), _F)
        xp[:, left:left + T] = x
        y = np.asarray(self.bias, _F)[None, None, :] + sum(xp[:, j:j + T] @ np.asarray(self.kernel[j]) for j in builtins_range(k))
        return _t(y)


class LayerNormalization(Layer):
    def __init__(self, epsilon=1e-3, name=None, **kw):
        super().__init__(name=name)
        self.epsilon = epsilon
        self.gamma = self.beta = None

    def call(self, x, training=None, **kw):
        x = np.asarray(x, _F)
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return _t((x - mu) / np.sqrt(var + self.epsilon) * np.asarray(self.gamma) + np.asarray(self.beta))


class BatchNormalization(Layer):
    def __init__(self, momentum=0.99, epsilon=1e-3, name=None, **kw):
        super().__init__(name=name)
        self.epsilon = epsilon
        self.gamma = self.beta = self.moving_mean = self.moving_variance = None

    def call(self, x, training=None, **kw):
        x = np.asarray(x, _F)
        training = _TRAINING[0] if training is None else training
        if training:
            mean, var = x.mean((0, 1)), x.var((0, 1))
        else:
            mean, var = np.asarray(self.moving_mean), np.asarray(self.moving_variance)
        return _t((x - mean) / np.sqrt(var + self.epsilon) * np.asarray(self.gamma) + np.asarray(self.beta))


class Dropout(Layer):
    def __init__(self, rate, name=None, **kw):
        super().__init__(name=name)
        self.rate = rate

    def call(self, x, training=None, **kw):
        training = _TRAINING[0] if training is None else training
        assert not training, "the shim runs inference-mode comparisons only"
        return x


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, name=None, **kw):
        super().__init__(name=name)
        self.embeddings = None

    def call(self, ids, **kw):
        return _t(np.asarray(self.embeddings)[np.asarray(ids)])


import builtins as _b
builtins_range = _b.range


class _Missing:
    def __init__(self, *a, **k):
        raise NotImplementedError("tf shim: symbol not on the text->mel path")


class _Layers(types.SimpleNamespace):
    def __getattr__(self, name):
        return _Missing


layers = _Layers(Layer=Layer, Dense=Dense, Conv1D=Conv1D, LayerNormalization=LayerNormalization,
                 BatchNormalization=BatchNormalization, Dropout=Dropout, Embedding=Embedding)
keras = types.SimpleNamespace(layers=layers, Model=Model,
                              initializers=types.SimpleNamespace(GlorotUniform=lambda *a, **k: None),
                              optimizers=types.SimpleNamespace())
