"""Weights contract of the VAENAR-TTS text->mel path.

The variable tree follows the reference's object-graph attribute paths
(models/models.py:16-65 and the module constructors: encoder.py:59-77,
attention.py:149-161,392-403,418-434, utils.py:21-31,41-46,56-74,98-109,
length_predictor.py:30-33, prior.py:79-99, flow.py:116-121,156-164,199-210,
transform.py:8-44, decoder.py:156-179, posterior.py:90-113).  A weight set is
a flat ``{path: float32 ndarray}`` dict, stored as one ``.npz``.

Kernel conventions (Keras, TF 2.2): Dense kernel ``[in, out]``; Conv1D kernel
``[k, in, out]``; Embedding ``[vocab, dim]``.
"""
from collections import OrderedDict

import numpy as np


def _xblk(spec, prefix, input_dim, att_dim, mem_dim, ffn):
    """CrossAttentionBLK variables (attention.py:418-434)."""
    for n in ("query", "key", "value"):
        spec[f"{prefix}/self_attention/{n}_layer/kernel"] = (input_dim, att_dim)
    spec[f"{prefix}/att_proj1/kernel"] = (input_dim + att_dim, input_dim)
    spec[f"{prefix}/att_proj1/bias"] = (input_dim,)
    spec[f"{prefix}/layer_norm1/gamma"] = (input_dim,)
    spec[f"{prefix}/layer_norm1/beta"] = (input_dim,)
    spec[f"{prefix}/cross_attention/query_layer/kernel"] = (input_dim, att_dim)
    spec[f"{prefix}/cross_attention/key_layer/kernel"] = (mem_dim, att_dim)
    spec[f"{prefix}/cross_attention/value_layer/kernel"] = (mem_dim, att_dim)
    spec[f"{prefix}/att_proj2/kernel"] = (2 * att_dim, att_dim)
    spec[f"{prefix}/att_proj2/bias"] = (att_dim,)
    spec[f"{prefix}/layer_norm2/gamma"] = (att_dim,)
    spec[f"{prefix}/layer_norm2/beta"] = (att_dim,)
    _ffn(spec, f"{prefix}/ffn", att_dim, ffn)


def _ffn(spec, prefix, dim, hidden):
    """FFN variables (utils.py:41-46)."""
    spec[f"{prefix}/dense1/kernel"] = (dim, hidden)
    spec[f"{prefix}/dense1/bias"] = (hidden,)
    spec[f"{prefix}/dense2/kernel"] = (hidden, dim)
    spec[f"{prefix}/dense2/bias"] = (dim,)
    spec[f"{prefix}/layer_norm/gamma"] = (dim,)
    spec[f"{prefix}/layer_norm/beta"] = (dim,)


def _conv(spec, prefix, k, cin, cout):
    """Conv1D + BatchNormalization variables (utils.py:56-74)."""
    spec[f"{prefix}/conv1d/kernel"] = (k, cin, cout)
    spec[f"{prefix}/conv1d/bias"] = (cout,)
    for n in ("gamma", "beta", "moving_mean", "moving_variance"):
        spec[f"{prefix}/bn/{n}"] = (cout,)


def weight_spec(hps, include_posterior=True):
    """Ordered ``{path: shape}`` for a hyper-parameter set."""
    s = OrderedDict()
    e = hps.Encoder.Transformer
    s["text_encoder/emb_layer/embeddings"] = (e.vocab_size, e.embd_dim)
    s["text_encoder/pos_weight"] = ()
    cin = e.embd_dim
    for i in range(e.n_conv):
        _conv(s, f"text_encoder/prenet/conv_stack/{i}", e.conv_kernel, cin, e.pre_hidden)
        cin = e.pre_hidden
    s["text_encoder/prenet/projection/kernel"] = (e.pre_hidden, e.pre_hidden)
    s["text_encoder/prenet/projection/bias"] = (e.pre_hidden,)
    for i in range(e.n_blk):
        p = f"text_encoder/self_attentions/{i}"
        for n in ("query", "key", "value"):
            s[f"{p}/attention/{n}_layer/kernel"] = (e.pre_hidden, e.attention_dim)
        s[f"{p}/att_proj/kernel"] = (e.pre_hidden + e.attention_dim, e.pre_hidden)
        s[f"{p}/att_proj/bias"] = (e.pre_hidden,)
        s[f"{p}/layer_norm/gamma"] = (e.pre_hidden,)
        s[f"{p}/layer_norm/beta"] = (e.pre_hidden,)
        _ffn(s, f"{p}/ffn", e.pre_hidden, e.ffn_hidden)
    mem = e.pre_hidden
    s["length_predictor/projection/kernel"] = (mem, 1)
    s["length_predictor/projection/bias"] = (1,)

    r = hps.Prior.Transformer
    C = hps.Common.latent_dim
    for st in range(r.n_blk):
        p = f"prior/glow/{st}"
        s[f"{p}/0/log_scale"] = (C,)
        s[f"{p}/0/bias"] = (C,)
        s[f"{p}/1/weight"] = (C, C)
        s[f"{p}/2/net/pos_weight"] = ()
        s[f"{p}/2/net/pre_projection/kernel"] = (C // 2, r.attention_dim)
        s[f"{p}/2/net/pre_projection/bias"] = (r.attention_dim,)
        for n in ("log_scale_proj", "shift_proj"):
            s[f"{p}/2/net/{n}/kernel"] = (r.attention_dim, C // 2)
            s[f"{p}/2/net/{n}/bias"] = (C // 2,)
        for b in range(r.n_transformer_blk):
            _xblk(s, f"{p}/2/net/attentions/{b}", r.attention_dim, r.attention_dim, mem,
                  r.ffn_hidden)

    d = hps.Decoder.Transformer
    out_dim = hps.Common.output_dim
    s["decoder/pre_projection/kernel"] = (C, d.attention_dim)
    s["decoder/pre_projection/bias"] = (d.attention_dim,)
    for b in range(d.nblk):
        _xblk(s, f"decoder/attentions/{b}", d.attention_dim, d.attention_dim, mem, d.ffn_hidden)
    s["decoder/out_projection/kernel"] = (d.attention_dim, out_dim * hps.Common.max_reduction_factor)
    s["decoder/out_projection/bias"] = (out_dim * hps.Common.max_reduction_factor,)
    cin = out_dim
    for i in range(d.post_n_conv):
        _conv(s, f"decoder/postnet/conv_stack/{i}", d.post_conv_kernel, cin, d.post_conv_filters)
        cin = d.post_conv_filters
    s["decoder/residual_projection/kernel"] = (d.post_conv_filters, out_dim)
    s["decoder/residual_projection/bias"] = (out_dim,)

    if include_posterior:
        q = hps.Posterior.Transformer
        s["posterior/pos_weight"] = ()
        s["posterior/prenet/dense1/kernel"] = (hps.Audio.num_mels, q.pre_hidden)
        s["posterior/prenet/dense1/bias"] = (q.pre_hidden,)
        s["posterior/prenet/dense2/kernel"] = (q.pre_hidden, q.pre_hidden)
        s["posterior/prenet/dense2/bias"] = (q.pre_hidden,)
        for b in range(q.nblk):
            _xblk(s, f"posterior/attentions/{b}", q.pre_hidden, q.attention_dim, mem, q.ffn_hidden)
        for n in ("mu_projection", "logvar_projection"):
            s[f"posterior/{n}/kernel"] = (q.attention_dim, C)
            s[f"posterior/{n}/bias"] = (C,)
    return s


def is_trainable(path):
    """BN moving statistics are the only non-trainable variables (train.py:136)."""
    return not (path.endswith("moving_mean") or path.endswith("moving_variance"))


def count_params(spec, trainable_only=False):
    return int(sum(int(np.prod(sh)) for p, sh in spec.items()
                   if not trainable_only or is_trainable(p)))


def init_weights(hps, seed=1234, mode="synthetic", include_posterior=True, dtype=np.float32):
    """Deterministic weight set.

    mode="reference": the reference initialisers — glorot-uniform kernels and
      zero biases (Keras defaults), Embedding U(-0.05, 0.05), pos_weight 1
      (encoder.py:64, transform.py:35, posterior.py:95), InvertibleLinear
      Q of qr(randn) (flow.py:120), ActNorm log_scale ~ N(0, 0.05), bias 0
      (flow.py:160-164), zero kernels for mu/logvar (posterior.py:108-113) and
      log_scale/shift projections (transform.py:12-17), BN gamma 1 / beta 0 /
      mean 0 / var 1, LN gamma 1 / beta 0.
    mode="synthetic": same, but every quantity that the reference initialises to
      a constant is perturbed so that no term of the arithmetic is trivially
      absent: zero-init heads -> N(0, 0.02), biases ->
      N(0, 0.02), LN/BN gamma -> 1 + N(0, 0.05), beta -> N(0, 0.05), BN
      moving_mean -> N(0, 0.1), moving_variance -> U(0.5, 1.5), ActNorm bias
      -> N(0, 0.1), pos_weight -> 1 + N(0, 0.05).
    """
    assert mode in ("reference", "synthetic")
    syn = mode == "synthetic"
    rng = np.random.Generator(np.random.PCG64(seed))
    w = OrderedDict()
    for path, shape in weight_spec(hps, include_posterior).items():
        leaf = path.rsplit("/", 1)[-1]
        if path.endswith("emb_layer/embeddings"):
            a = rng.uniform(-0.05, 0.05, shape)
        elif leaf == "pos_weight":
            a = np.asarray(1.0 + (rng.normal(0, 0.05) if syn else 0.0))
        elif path.endswith("/1/weight"):
            a = np.linalg.qr(rng.standard_normal(shape))[0]
        elif path.endswith("/0/log_scale"):
            a = rng.normal(0, 0.05, shape)
        elif path.endswith("/0/bias"):
            a = rng.normal(0, 0.1, shape) if syn else np.zeros(shape)
        elif leaf == "kernel":
            zero_init = any(t in path for t in ("log_scale_proj", "shift_proj",
                                                "mu_projection", "logvar_projection"))
            if zero_init:
                a = rng.normal(0, 0.02, shape) if syn else np.zeros(shape)
            else:
                # glorot uniform: fan_in/out include the receptive field for conv
                rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
                fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                a = rng.uniform(-lim, lim, shape)
        elif leaf == "bias":
            a = rng.normal(0, 0.02, shape) if syn else np.zeros(shape)
        elif leaf == "gamma":
            a = 1.0 + (rng.normal(0, 0.05, shape) if syn else np.zeros(shape))
        elif leaf == "beta":
            a = rng.normal(0, 0.05, shape) if syn else np.zeros(shape)
        elif leaf == "moving_mean":
            a = rng.normal(0, 0.1, shape) if syn else np.zeros(shape)
        elif leaf == "moving_variance":
            a = rng.uniform(0.5, 1.5, shape) if syn else np.ones(shape)
        else:
            raise KeyError(path)
        w[path] = np.ascontiguousarray(a, dtype=dtype).reshape(shape)      # (ascontiguousarray promotes 0-d to 1-d)
    return w


def save_npz(path, weights):
    np.savez(path, **{k.replace("/", "|"): v for k, v in weights.items()})


def load_npz(path):
    with np.load(path) as z:
        return OrderedDict((k.replace("|", "/"), z[k]) for k in z.files)
