"""A `tensorflow`-named shim over torch CPU float64 tensors WITH autograd -- TEST INFRASTRUCTURE, this container only.

Purpose (VERDICT round 1, item 4 / SURVEY section 8 C'(ii)): let the reference's own ``models/models.py`` and ``modules/*.py``
(which ``import tensorflow as tf``) execute in TRAINING mode -- ``model(..., training=True)``, ``model.init(...)`` -- and let
``torch.autograd`` differentiate the loss of train.py:135 with respect to every variable, so that the fixtures
``tests/golden/refshim_train_*.npz`` are produced by the reference's Python, not by this repository's restatements.

Same scope and semantics as oracle/tf_shim (the ~75 tf symbols of the text->mel path, TF 2.2 / Keras documented behaviour,
SURVEY.md Appendix A; "float32" maps to float64), plus what training needs:
  * Dropout draws its keep-mask from an injectable source: every Dropout INSTANCE carries the list of engine "sites" it
    stands for (oracle/run_reference_on_shim.assign_dropout_sites) and the mask of call #i of an instance is
    oracle.vaenar_numpy.dropout_keep(shape, rate, SEED, site_i) -- the counter-based mask the engine draws; kept values are
    scaled by 1 / (1 - rate) (tf.nn.dropout);
  * BatchNormalization in training mode normalises with the batch mean / population variance over axes (0, 1) and updates the
    moving statistics with momentum 0.99 (non-fused Keras path for rank-3 inputs: no Bessel correction);
  * Keras `training` call-context propagation (a sub-layer called without `training=` inherits the caller's value);
  * tf.Variable = a leaf tensor that requires grad; `.assign` writes in place (ActNorm data-dependent init, BN statistics).
It does NOT pin TensorFlow's kernel numerics: parity with the real reference stays unpinned (oracle/vaenar_numpy.py header).
Never shipped to the GPU box, never imported by vaenar_tts_amd.
"""
import builtins as _b
import types

import numpy as np
import torch

_F = torch.float64
float32 = "float32"
float64 = "float64"
int32 = "int32"
int64 = "int64"
bool = "bool"  # noqa: A001  (tf.bool)


def _dt(d):
    if d is None:
        return None
    if d in ("float32", "float64") or d is float or d in (torch.float32, torch.float64, np.float32, np.float64):
        return _F
    if d in ("int32", "int64") or d is int or d in (torch.int32, torch.int64, np.int32, np.int64):
        return torch.int64
    if d in ("bool",) or d is _b.bool or d is torch.bool:
        return torch.bool
    raise TypeError("tf shim: dtype %r" % (d,))


class Tensor(torch.Tensor):
    """torch.Tensor with the handful of tf.Tensor methods the reference calls."""

    def set_shape(self, shape):
        return None

    def numpy(self):
        return torch.Tensor.numpy(self.detach().as_subclass(torch.Tensor))

    def assign(self, v):
        with torch.no_grad():
            self.copy_(_raw(v).to(self.dtype))
        return self

    def assign_add(self, v):
        with torch.no_grad():
            self.add_(_raw(v).to(self.dtype))
        return self

    def __bool__(self):
        return _b.bool(torch.Tensor.__bool__(self.detach().as_subclass(torch.Tensor)))

    def __index__(self):
        return int(self.item())

    def __hash__(self):
        return id(self)


def _raw(x, dtype=None):
    """anything -> torch tensor (float data in float64)"""
    if isinstance(x, torch.Tensor):
        t = x
    else:
        a = np.asarray(x)
        t = torch.as_tensor(a.astype(np.float64) if a.dtype.kind == "f" else (a.astype(np.int64) if a.dtype.kind in "iu" else a))
    if dtype is not None:
        t = t.to(dtype)
    elif t.dtype in (torch.float32, torch.float16):
        t = t.to(_F)
    return t


def _t(x):
    t = _raw(x)
    return t if isinstance(t, Tensor) else t.as_subclass(Tensor)


def _ints(s):
    if isinstance(s, torch.Tensor):
        return [int(i) for i in s.reshape(-1).tolist()]
    if isinstance(s, (list, tuple)):
        return [int(i) for i in s]
    return [int(i) for i in np.asarray(s).reshape(-1)]


TensorShape = list


def Variable(initial_value, trainable=True, dtype=None, name=None):
    t = _raw(initial_value).detach().clone()
    if t.dtype.is_floating_point:
        t = t.to(_F)
    v = t.as_subclass(Tensor)
    v.requires_grad_(_b.bool(trainable) and t.dtype.is_floating_point)
    v.trainable = trainable
    v.is_variable = True
    return v


def _is_variable(x):
    return isinstance(x, torch.Tensor) and getattr(x, "is_variable", False)


def constant(v, dtype=None, shape=None, name=None):
    return _t(_raw(v, _dt(dtype)))


def shape(x):
    return _t(torch.tensor(list(_raw(x).shape), dtype=torch.int64))


def reshape(x, s):
    return _t(_raw(x).reshape(_ints(s)))


def transpose(x, perm=None):
    x = _raw(x)
    return _t(x.permute(*_ints(perm)) if perm is not None else x.t())


def tile(x, m):
    return _t(_raw(x).repeat(*_ints(m)))


def expand_dims(x, axis):
    return _t(_raw(x).unsqueeze(int(axis)))


def concat(xs, axis):
    return _t(torch.cat([_raw(x) for x in xs], int(axis)))


def split(x, num_or_size_splits, axis=0):
    x = _raw(x)
    n = int(num_or_size_splits)
    return [_t(p) for p in torch.chunk(x, n, int(axis))]


def cast(x, dtype):
    d = _dt(dtype)
    x = _raw(x)
    if d is torch.int64 and x.dtype.is_floating_point:
        x = torch.trunc(x)                     # tf.cast(float -> int32) truncates toward zero
    return _t(x.to(d))


def range(start, limit=None, delta=1, dtype=None):  # noqa: A001
    if limit is None:
        start, limit = 0, start
    d = _dt(dtype)
    f = lambda v: float(v) if d is _F else int(v)                     # noqa: E731
    return _t(torch.arange(f(start), f(limit), f(delta), dtype=d or torch.int64))


def ones(s, dtype=float32):
    return _t(torch.ones(_ints(s), dtype=_dt(dtype)))


def zeros(s, dtype=float32):
    return _t(torch.zeros(_ints(s), dtype=_dt(dtype)))


def ones_like(x, dtype=None):
    x = _raw(x)
    return _t(torch.ones(x.shape, dtype=_dt(dtype) or x.dtype))


def where(c, x=None, y=None):
    x, y = _raw(x), _raw(y)
    return _t(torch.where(_raw(c).to(torch.bool), x, y.to(x.dtype)))


def sequence_mask(lengths, maxlen=None, dtype=bool, name=None):
    lengths = _raw(lengths).to(torch.int64)
    if maxlen is None:
        maxlen = int(lengths.max())
    m = torch.arange(int(maxlen))[None, :] < lengths[..., None]
    return _t(m.to(_dt(dtype)))


def stop_gradient(x):
    return _t(_raw(x).detach())


def identity(x):
    return x


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    a, b = _raw(a), _raw(b)
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return _t(a @ b)


def _axis(axis):
    if axis is None:
        return None
    return tuple(int(i) for i in axis) if isinstance(axis, (list, tuple)) else int(axis)


def reduce_sum(x, axis=None, keepdims=False):
    x = _raw(x)
    return _t(x.sum() if axis is None else x.sum(_axis(axis), keepdim=keepdims))


def reduce_mean(x, axis=None, keepdims=False):
    x = _raw(x)
    return _t(x.mean() if axis is None else x.mean(_axis(axis), keepdim=keepdims))


def reduce_max(x, axis=None):
    x = _raw(x)
    return _t(x.max() if axis is None else x.amax(_axis(axis)))


def _reduce_std(x, axis=None):
    x = _raw(x)                                  # population std (ddof 0), tf.math.reduce_std
    mu = x.mean() if axis is None else x.mean(_axis(axis), keepdim=True)
    v = ((x - mu) ** 2)
    return _t(torch.sqrt(v.mean() if axis is None else v.mean(_axis(axis))))


def exp(x):
    return _t(torch.exp(_raw(x, _F)))


def sqrt(x):
    return _t(torch.sqrt(_raw(x, _F)))


def square(x):
    return _t(_raw(x) ** 2)


def pow(x, y):  # noqa: A001
    return _t(torch.pow(_raw(x, _F), _raw(y, _F)))


def abs(x):  # noqa: A001
    return _t(torch.abs(_raw(x)))


def maximum(a, b):
    a = _raw(a)
    return _t(torch.maximum(a, _raw(b).to(a.dtype)))


def logical_and(a, b):
    return _t(torch.logical_and(_raw(a), _raw(b)))


def stack(xs, axis=0):
    return _t(torch.stack([_raw(x) for x in xs], int(axis)))


def _sigmoid(x):
    x = _raw(x, _F)
    return _t(1.0 / (1.0 + torch.exp(-x)))


def _softmax(x, axis=-1):
    x = _raw(x, _F)                              # tf.math.softmax: exp(x - max) / sum
    e = torch.exp(x - x.amax(int(axis), keepdim=True))
    return _t(e / e.sum(int(axis), keepdim=True))


math = types.SimpleNamespace(
    softmax=_softmax, sigmoid=_sigmoid, log=lambda x: _t(torch.log(_raw(x, _F))), exp=exp, sqrt=sqrt,
    sin=lambda x: _t(torch.sin(_raw(x, _F))), cos=lambda x: _t(torch.cos(_raw(x, _F))),
    mod=lambda a, b: _t(torch.remainder(_raw(a), b)), equal=lambda a, b: _t(torch.eq(_raw(a), _raw(b) if isinstance(b, torch.Tensor) else b)),
    logical_and=logical_and, reduce_sum=reduce_sum, reduce_mean=reduce_mean, reduce_max=reduce_max, reduce_std=_reduce_std,
    tanh=lambda x: _t(torch.tanh(_raw(x, _F))), maximum=maximum)
nn = types.SimpleNamespace(relu=lambda x: _t(torch.relu(_raw(x))), tanh=math.tanh, sigmoid=_sigmoid)


def _band_part(x, lower, upper, name=None):
    a = _raw(x)
    r, c = a.shape[-2:]
    i, j = torch.arange(r)[:, None], torch.arange(c)[None, :]
    keep = torch.ones((r, c), dtype=torch.bool)
    if lower >= 0:
        keep &= (i - j) <= lower
    if upper >= 0:
        keep &= (j - i) <= upper
    return _t(torch.where(keep, a, torch.zeros_like(a)))


def _slogdet(a):
    s, l = torch.linalg.slogdet(_raw(a, _F))
    return _t(s), _t(l)


linalg = types.SimpleNamespace(matmul=matmul, band_part=_band_part, slogdet=_slogdet, inv=lambda a: _t(torch.linalg.inv(_raw(a, _F))))


class _Random:
    """tf.random.normal with an injectable queue: the generator script pushes the noise the engine is given."""
    queue = []

    @classmethod
    def normal(cls, shape, mean=0.0, stddev=1.0, dtype=None):
        shp = tuple(_ints(shape))
        mean, stddev = float(_raw(mean)), float(_raw(stddev))
        if cls.queue:
            a = _raw(cls.queue.pop(0), _F)
            assert tuple(a.shape) == shp, (tuple(a.shape), shp)
            return _t(a * stddev + mean)
        return _t(torch.zeros(shp, dtype=_F) + mean)


random = _Random
nest = types.SimpleNamespace(flatten=lambda x: list(x) if isinstance(x, (list, tuple)) else [x])
losses = types.SimpleNamespace(MeanSquaredError=object, MeanAbsoluteError=object)


# ---- tf.keras ---------------------------------------------------------------------------------------------------------
_TRAINING = [None]        # Keras call-context propagation of `training`
DROPOUT = {"seed": 0}     # seed of the counter-based masks (oracle.vaenar_numpy.dropout_keep)


class Layer:
    def __init__(self, name=None, **kwargs):
        self.name = name

    def __call__(self, *args, **kwargs):
        outer = _TRAINING[0]
        if kwargs.get("training", None) is not None:
            _TRAINING[0] = _b.bool(kwargs["training"])
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
        y = _raw(x, _F) @ self.kernel
        if self.use_bias:
            y = y + self.bias
        y = _t(y)
        return self.activation(y) if self.activation is not None else y


class Conv1D(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding="valid", activation=None, name=None, **kw):
        super().__init__(name=name)
        assert strides == 1 and str(padding).lower() == "same" and activation is None
        self.kernel = self.bias = None

    def call(self, x, **kw):
        x = _raw(x, _F)                                  # [B, T, Cin]; kernel [k, Cin, Cout]; cross-correlation, 'same' zero padding
        k = self.kernel.shape[0]
        left = (k - 1) // 2
        xp = torch.nn.functional.pad(x, (0, 0, left, k - 1 - left))
        T = x.shape[1]
        y = self.bias[None, None, :]
        for j in _b.range(k):
            y = y + xp[:, j:j + T] @ self.kernel[j]
        return _t(y)


class LayerNormalization(Layer):
    def __init__(self, epsilon=1e-3, name=None, **kw):
        super().__init__(name=name)
        self.epsilon = epsilon
        self.gamma = self.beta = None

    def call(self, x, training=None, **kw):
        x = _raw(x, _F)
        mu = x.mean(-1, keepdim=True)
        var = ((x - mu) ** 2).mean(-1, keepdim=True)
        return _t((x - mu) / torch.sqrt(var + self.epsilon) * self.gamma + self.beta)


class BatchNormalization(Layer):
    def __init__(self, momentum=0.99, epsilon=1e-3, name=None, **kw):
        super().__init__(name=name)
        self.epsilon, self.momentum = epsilon, momentum
        self.gamma = self.beta = self.moving_mean = self.moving_variance = None

    def call(self, x, training=None, **kw):
        x = _raw(x, _F)
        training = _TRAINING[0] if training is None else _b.bool(training)
        if training:
            mean = x.mean((0, 1))
            var = ((x - mean) ** 2).mean((0, 1))          # population variance (non-fused Keras path, rank-3 input)
            with torch.no_grad():                        # moving <- moving * momentum + batch * (1 - momentum)
                self.moving_mean.mul_(self.momentum).add_(mean.detach() * (1.0 - self.momentum))
                self.moving_variance.mul_(self.momentum).add_(var.detach() * (1.0 - self.momentum))
        else:
            mean, var = self.moving_mean, self.moving_variance
        return _t((x - mean) / torch.sqrt(var + self.epsilon) * self.gamma + self.beta)


class Dropout(Layer):
    def __init__(self, rate, name=None, **kw):
        super().__init__(name=name)
        self.rate = float(rate)
        self.sites = None          # engine sites of call #0, #1, ... of this instance within one forward (assign_dropout_sites)
        self.calls = 0

    def call(self, x, training=None, **kw):
        training = _TRAINING[0] if training is None else _b.bool(training)
        if not training or self.rate <= 0.0:
            return x
        from oracle.vaenar_numpy import dropout_keep
        assert self.sites is not None and self.calls < len(self.sites), "Dropout instance without an assigned engine site"
        x = _raw(x, _F)
        keep = dropout_keep(tuple(x.shape), self.rate, DROPOUT["seed"], self.sites[self.calls])
        self.calls += 1
        # tf.nn.dropout: kept values scaled by 1 / (1 - rate); the rate is a float32 in the engine and in hparams
        scale = 1.0 / (1.0 - float(np.float32(self.rate)))
        return _t(x * torch.as_tensor(keep, dtype=_F) * scale)


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, name=None, **kw):
        super().__init__(name=name)
        self.embeddings = None

    def call(self, ids, **kw):
        return _t(self.embeddings[_raw(ids).to(torch.int64)])


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
