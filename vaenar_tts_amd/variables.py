"""`model.trainable_variables` of the reference (train.py:136-137) on the MI355X engine.

The reference hands `tape.gradient(loss, model.trainable_variables)` and
`optimizer.apply_gradients(zip(gradients, model.trainable_variables))` the Keras variable list of `models.VAENAR`.  Here a
`Variable` is a view of one tensor of the engine's weight store (C ABI: vnr_get_weight / vnr_set_weight / vnr_get_gradient /
vnr_get_optimizer_slot); the list order is the one Keras produces for the reference's constructors: a layer's own
`tf.Variable`s first, then its sub-layers in the order the constructor attached them (a Dense / Conv1D contributes kernel, bias;
LayerNormalization / BatchNormalization gamma, beta; the BatchNormalization moving statistics are the only non-trainable
variables).
"""
import numpy as np

from .weights import is_trainable, weight_spec


def _ffn(p):                                            # utils.py:41-46
    return [f"{p}/dense1/kernel", f"{p}/dense1/bias", f"{p}/dense2/kernel", f"{p}/dense2/bias",
            f"{p}/layer_norm/gamma", f"{p}/layer_norm/beta"]


def _mha(p):                                            # attention.py:149-161
    return [f"{p}/query_layer/kernel", f"{p}/key_layer/kernel", f"{p}/value_layer/kernel"]


def _xblk(p):                                           # attention.py:418-434
    return (_mha(f"{p}/self_attention") + [f"{p}/att_proj1/kernel", f"{p}/att_proj1/bias",
                                           f"{p}/layer_norm1/gamma", f"{p}/layer_norm1/beta"]
            + _mha(f"{p}/cross_attention") + [f"{p}/att_proj2/kernel", f"{p}/att_proj2/bias",
                                              f"{p}/layer_norm2/gamma", f"{p}/layer_norm2/beta"] + _ffn(f"{p}/ffn"))


def _conv(p, trainable=True):                           # utils.py:56-74
    if trainable:
        return [f"{p}/conv1d/kernel", f"{p}/conv1d/bias", f"{p}/bn/gamma", f"{p}/bn/beta"]
    return [f"{p}/bn/moving_mean", f"{p}/bn/moving_variance"]


def keras_variable_order(hps, trainable_only=True, include_posterior=True):
    """Object-graph paths of the model's variables in the order of `model.trainable_variables` (`trainable_only`) or of
    `model.variables` = trainable_variables + non_trainable_variables (the BatchNormalization moving statistics, layer order)."""
    t = True
    e, d, q, r = hps.Encoder.Transformer, hps.Decoder.Transformer, hps.Posterior.Transformer, hps.Prior.Transformer
    o = []
    # models.py:16-30 text_encoder (encoder.py:8-13,59-77): own pos_weight, then emb_layer, prenet, self_attentions
    o += ["text_encoder/pos_weight", "text_encoder/emb_layer/embeddings"]
    for i in range(e.n_conv):                            # utils.py:21-31
        o += _conv(f"text_encoder/prenet/conv_stack/{i}", t)
    o += ["text_encoder/prenet/projection/kernel", "text_encoder/prenet/projection/bias"]
    for i in range(e.n_blk):                             # attention.py:392-403
        p = f"text_encoder/self_attentions/{i}"
        o += _mha(f"{p}/attention") + [f"{p}/att_proj/kernel", f"{p}/att_proj/bias", f"{p}/layer_norm/gamma",
                                       f"{p}/layer_norm/beta"] + _ffn(f"{p}/ffn")
    # models.py:31-43 decoder (decoder.py:156-179)
    o += ["decoder/pre_projection/kernel", "decoder/pre_projection/bias"]
    for b in range(d.nblk):
        o += _xblk(f"decoder/attentions/{b}")
    o += ["decoder/out_projection/kernel", "decoder/out_projection/bias"]
    for i in range(d.post_n_conv):                       # utils.py:98-109
        o += _conv(f"decoder/postnet/conv_stack/{i}", t)
    o += ["decoder/residual_projection/kernel", "decoder/residual_projection/bias"]
    # models.py:44-45 length_predictor (length_predictor.py:30-33)
    o += ["length_predictor/projection/kernel", "length_predictor/projection/bias"]
    # models.py:46-56 posterior (posterior.py:90-113): own pos_weight, prenet, attentions, mu / logvar projections
    if include_posterior:
        o += ["posterior/pos_weight", "posterior/prenet/dense1/kernel", "posterior/prenet/dense1/bias",
              "posterior/prenet/dense2/kernel", "posterior/prenet/dense2/bias"]
        for b in range(q.nblk):
            o += _xblk(f"posterior/attentions/{b}")
        o += ["posterior/mu_projection/kernel", "posterior/mu_projection/bias",
              "posterior/logvar_projection/kernel", "posterior/logvar_projection/bias"]
    # models.py:57-65 prior (prior.py:79-99): per flow step (actnorm, linear, coupling); flow.py:156-164,116-121,199-210;
    # the coupling's net (transform.py:8-44): own pos_weight, then log_scale_proj, shift_proj (BaseTransform), pre_projection, attentions
    for s in range(r.n_blk):
        p = f"prior/glow/{s}"
        o += [f"{p}/0/log_scale", f"{p}/0/bias", f"{p}/1/weight", f"{p}/2/net/pos_weight",
              f"{p}/2/net/log_scale_proj/kernel", f"{p}/2/net/log_scale_proj/bias",
              f"{p}/2/net/shift_proj/kernel", f"{p}/2/net/shift_proj/bias",
              f"{p}/2/net/pre_projection/kernel", f"{p}/2/net/pre_projection/bias"]
        for b in range(r.n_transformer_blk):
            o += _xblk(f"{p}/2/net/attentions/{b}")
    if not trainable_only:
        for i in range(e.n_conv):
            o += _conv(f"text_encoder/prenet/conv_stack/{i}", False)
        for i in range(d.post_n_conv):
            o += _conv(f"decoder/postnet/conv_stack/{i}", False)
    spec = weight_spec(hps, include_posterior)
    assert sorted(o) == sorted(k for k in spec if (is_trainable(k) or not trainable_only)), "variable order does not cover the weights contract"
    return o


class Variable:
    """One variable of the model (tf.Variable's role): `.name`, `.shape`, `.trainable`, `.numpy()`, `.assign(value)`;
    `.gradient()` / `.slot("m" | "v")` read d loss / d variable of the last train_step and Adam's moments."""

    __slots__ = ("engine", "path", "shape", "trainable", "dtype")

    def __init__(self, engine, path, shape, trainable):
        self.engine, self.path, self.shape, self.trainable = engine, path, tuple(shape), trainable
        self.dtype = np.dtype(np.float32)

    @property
    def name(self):
        return self.path + ":0"

    def _flat(self):
        return self.shape if len(self.shape) else (1,)

    def numpy(self):
        return self.engine.get_weight(self.path, self._flat()).reshape(self.shape)

    def assign(self, value):
        """tf.Variable.assign: the engine re-packs its kernel panels lazily before the next module call."""
        a = np.asarray(value.numpy() if hasattr(value, "numpy") and not isinstance(value, np.ndarray) else value, np.float32)
        if a.shape != self.shape:
            raise ValueError("%s: cannot assign shape %s to a variable of shape %s" % (self.path, a.shape, self.shape))
        self.engine.set_weight(self.path, a.reshape(self.shape))
        return self

    def assign_sub(self, delta):
        return self.assign(self.numpy() - np.asarray(delta, np.float32))

    def assign_add(self, delta):
        return self.assign(self.numpy() + np.asarray(delta, np.float32))

    def gradient(self):
        if not self.trainable:
            raise ValueError("%s is not trainable (BatchNormalization moving statistic)" % self.path)
        return self.engine.get_gradient(self.path, self._flat()).reshape(self.shape)

    def slot(self, name):
        return self.engine.get_optimizer_slot(self.path, name, self._flat()).reshape(self.shape)

    def __repr__(self):
        return "<Variable %s shape=%s trainable=%s>" % (self.name, self.shape, self.trainable)


def model_variables(engine, hps, trainable_only=True, prefix=None, include_posterior=True):
    spec = weight_spec(hps, include_posterior)
    return [Variable(engine, p, spec[p], is_trainable(p))
            for p in keras_variable_order(hps, trainable_only, include_posterior)
            if prefix is None or p.startswith(prefix + "/")]
