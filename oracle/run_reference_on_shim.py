#!/usr/bin/env python3
"""Drive the REFERENCE's own Python (models.VAENAR + modules.*) over oracle/tf_shim and return its outputs.
This container only (/root/reference is not present on the GPU box)."""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_reference(backend="numpy"):
    """Import the reference packages with `tensorflow` resolved to a shim: ``numpy`` = oracle/tf_shim (inference-mode
    composition), ``torch`` = oracle/tf_shim_torch (float64 torch tensors with autograd: training mode, init, gradients)."""
    for m in [k for k in sys.modules if k == "tensorflow" or k.split(".")[0] in ("modules", "models", "configs")]:
        del sys.modules[m]
    shim = os.path.join(HERE, "tf_shim_torch" if backend == "torch" else "tf_shim")
    sys.path.insert(0, REF)
    sys.path.insert(0, shim)
    try:
        tf = importlib.import_module("tensorflow")
        assert tf.__file__.startswith(HERE), "a real tensorflow is installed; the shim is not needed"
        models = importlib.import_module("models.models")
        hp = importlib.import_module("configs.hparams")
    finally:
        sys.path.remove(REF)
        sys.path.remove(shim)
    return tf, models, hp


def _resolve(obj, path):
    for part in path:
        obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
    return obj


def set_weights(model, weights):
    """Assign {object-graph path: array} onto the reference model's layers (tf.train.Checkpoint naming)."""
    import tensorflow as tf
    for path, arr in weights.items():
        parts = path.split("/")
        owner, leaf = _resolve(model, parts[:-1]), parts[-1]
        val = np.asarray(arr, np.float64)
        cur = getattr(owner, leaf, None)
        if isinstance(cur, tf.Variable):          # pos_weight, ActNorm/InvertibleLinear variables
            assert cur.shape == val.shape, (path, cur.shape, val.shape)
            setattr(owner, leaf, tf.Variable(val))
        else:
            setattr(owner, leaf, val)


def apply_overrides(hps_cls, ours):
    """Copy the (possibly reduced) hyper-parameters of our config object onto the reference's LJHPS class."""
    def rec(dst, src):
        for k, v in src.__dict__.items():
            if hasattr(v, "__dict__") and not isinstance(v, (list, str)):
                rec(getattr(dst, k), v)
            elif isinstance(v, str) and v in ("relu", "tanh", "identity"):
                pass                                  # activations stay the reference's tf callables
            else:
                setattr(dst, k, v)
    rec(hps_cls, ours)


def reference_inference(ours_hps, weights, ids, mel_lengths, text_lengths, eps, reduction_factor=2):
    tf, models, hp = load_reference()
    apply_overrides(hp.LJHPS, ours_hps)
    model = models.VAENAR(hp.LJHPS)
    set_weights(model, weights)
    tf.random.queue[:] = [np.asarray(eps, np.float64)]
    mel, ali = model.inference(tf.constant(ids), tf.constant(mel_lengths), tf.constant(text_lengths), reduction_factor)
    return np.asarray(mel), {k: np.asarray(v) for k, v in ali.items()}, model, tf


def reference_call(ours_hps, weights, ids, mels, mel_lengths, text_lengths, eps, reduction_factor=2):
    tf, models, hp = load_reference()
    apply_overrides(hp.LJHPS, ours_hps)
    model = models.VAENAR(hp.LJHPS)
    set_weights(model, weights)
    tf.random.queue[:] = [np.asarray(eps, np.float64)]
    out = model(tf.constant(ids), tf.constant(np.asarray(mels, np.float64)), tf.constant(mel_lengths),
                tf.constant(text_lengths), reduction_factor=reduction_factor, training=False, reduce_loss=False)
    return out


# ---- training mode on the torch-backed shim: the reference's own forward under autograd ------------------------------------
# engine dropout sites (csrc/engine.hip: SITE_*), one per tf.keras.layers.Dropout CALL of the path
SITE_ENC_CONV, SITE_ENC_PE, SITE_POST_PRENET1, SITE_POST_PRENET2, SITE_POST_PE, SITE_POSTNET_CONV = 0, 8, 16, 17, 18, 32


def set_weights_torch(model, weights):
    """Assign {object-graph path: array} as torch leaves (tf.Variable of the torch shim); returns {path: tensor}."""
    import tensorflow as tf
    from vaenar_tts_amd.weights import is_trainable
    leaves = {}
    for path, arr in weights.items():
        parts = path.split("/")
        owner, leaf = _resolve(model, parts[:-1]), parts[-1]
        v = tf.Variable(np.asarray(arr, np.float64), trainable=is_trainable(path))
        setattr(owner, leaf, v)
        leaves[path] = v
    return leaves


def assign_dropout_sites(model):
    """Tell every Dropout instance of the reference model which engine mask stream(s) it stands for (call order within one
    forward): encoder.py:70,87; utils.py:11-17 (ONE layer called twice); posterior.py:99,122; utils.py:73,84."""
    for i, conv in enumerate(model.text_encoder.prenet.conv_stack):
        conv.dropout.sites = [SITE_ENC_CONV + i]
    model.text_encoder.pe_dropout.sites = [SITE_ENC_PE]
    model.posterior.prenet.dropout_layer.sites = [SITE_POST_PRENET1, SITE_POST_PRENET2]
    model.posterior.pe_dropout.sites = [SITE_POST_PE]
    for i, conv in enumerate(model.decoder.postnet.conv_stack):
        conv.dropout.sites = [SITE_POSTNET_CONV + i]


def _reset_dropout_calls(model):
    layers = [c.dropout for c in model.text_encoder.prenet.conv_stack] + [model.text_encoder.pe_dropout,
              model.posterior.prenet.dropout_layer, model.posterior.pe_dropout] + [c.dropout for c in model.decoder.postnet.conv_stack]
    for l in layers:
        l.calls = 0


def _eps4(eps):
    """[B, Tz, C] (n_sample = 1) or [B, n_sample, Tz, C] -> the [batch, nsamples, time, dim] tensor posterior.py:35 draws."""
    e = np.asarray(eps, np.float64)
    return e[:, None] if e.ndim == 3 else e


def _build(ours_hps, weights):
    tf, models, hp = load_reference("torch")
    apply_overrides(hp.LJHPS, ours_hps)
    model = models.VAENAR(hp.LJHPS)
    leaves = set_weights_torch(model, weights)
    assign_dropout_sites(model)
    return tf, model, leaves


def reference_train_step(ours_hps, weights, ids, mels, mel_lengths, text_lengths, eps, reduction_factor, kl_weight, dropout_seed):
    """train_step of train.py:127-138 up to the gradients, executed by the REFERENCE's model code: model(training=True,
    reduce_loss=True), loss of train.py:135, d loss / d every trainable variable (torch.autograd in place of tape.gradient).
    Returns (scalars, gradients {path: ndarray}, predictions, moving statistics after the forward {path: ndarray})."""
    import torch
    tf, model, leaves = _build(ours_hps, weights)
    tf.DROPOUT["seed"] = int(dropout_seed)
    _reset_dropout_calls(model)
    tf.random.queue[:] = [_eps4(eps)]                                                # posterior.py:35: [batch, nsamples, time, dim]
    preds, mel_l2, kl, length_l2, _ = model(inputs=tf.constant(ids), mel_targets=tf.constant(np.asarray(mels, np.float64)),
                                            mel_lengths=tf.constant(mel_lengths), text_lengths=tf.constant(text_lengths),
                                            reduction_factor=reduction_factor, training=True, reduce_loss=True)
    loss = mel_l2 + kl_weight * tf.math.maximum(kl, 0.) + ours_hps.Train.length_weight * length_l2         # train.py:135
    names = [k for k, v in leaves.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    g = {k: (np.zeros(tuple(leaves[k].shape)) if x is None else x.numpy().copy()) for k, x in zip(names, grads)}
    stats = {k: v.numpy().copy() for k, v in leaves.items() if k.endswith("moving_mean") or k.endswith("moving_variance")}
    sc = dict(loss=float(loss), mel_l2=float(mel_l2), kl=float(kl), length_l2=float(length_l2))
    return sc, g, preds.numpy().copy(), stats


def reference_call_training(ours_hps, weights, ids, mels, mel_lengths, text_lengths, eps, reduction_factor, dropout_seed):
    """VAENAR.call(training=True, reduce_loss=False) (models.py:105-197) by the reference's code: per-utterance terms."""
    tf, model, leaves = _build(ours_hps, weights)
    tf.DROPOUT["seed"] = int(dropout_seed)
    _reset_dropout_calls(model)
    tf.random.queue[:] = [_eps4(eps)]
    outs, l2, kl, ll, ali = model(inputs=tf.constant(ids), mel_targets=tf.constant(np.asarray(mels, np.float64)),
                                  mel_lengths=tf.constant(mel_lengths), text_lengths=tf.constant(text_lengths),
                                  reduction_factor=reduction_factor, training=True, reduce_loss=False)
    return outs.numpy().copy(), l2.numpy().copy(), kl.numpy().copy(), ll.numpy().copy(), {k: v.numpy().copy() for k, v in ali.items()}


def reference_init(ours_hps, weights, ids, mel_lengths, text_lengths, eps, dropout_seed):
    """VAENAR.init (models.py:212-226): the reference's data-dependent ActNorm initialisation; returns the predicted mel and
    every variable afterwards (ActNorm log_scale / bias, BN moving statistics)."""
    tf, model, leaves = _build(ours_hps, weights)
    tf.DROPOUT["seed"] = int(dropout_seed)
    _reset_dropout_calls(model)
    tf.random.queue[:] = [np.asarray(eps, np.float64)]
    mel = model.init(text_inputs=tf.constant(ids), mel_lengths=tf.constant(mel_lengths), text_lengths=tf.constant(text_lengths))
    return mel.numpy().copy(), {k: v.numpy().copy() for k, v in leaves.items()}


def reference_module_methods(ours_hps, weights, ids, text_lengths, mels_reduced, z_lengths, eps_post, eps_prior, eps_init, dropout_seed):
    """The module-level methods VAENAR.call / VAENAR.init reach into (models.py:141-144,216-219), executed by the REFERENCE's own
    classes: BasePosterior.reparameterize / log_probability (posterior.py:21-72) with nsamples = eps_post.shape[1],
    TransformerPrior.call / sample / log_probability with training=True and training=False (prior.py:101-169: the flag reaches no
    training-dependent layer), TransformerPrior.init (prior.py:171-186) with the variables it assigns."""
    tf, model, leaves = _build(ours_hps, weights)
    tf.DROPOUT["seed"] = int(dropout_seed)
    _reset_dropout_calls(model)
    out = {}
    tl, zl = tf.constant(text_lengths), tf.constant(z_lengths)
    text = model.text_encoder(tf.constant(ids), tl, pos_step=1.0, training=False)
    out["text_embd"] = text.numpy().copy()
    mu, logvar, _ = model.posterior(tf.constant(np.asarray(mels_reduced, np.float64)), text, src_lengths=tl, target_lengths=zl, training=False)
    out["mu"], out["logvar"] = mu.numpy().copy(), logvar.numpy().copy()
    ns = int(np.asarray(eps_post).shape[1])
    tf.random.queue[:] = [np.asarray(eps_post, np.float64)]
    samples, eps = model.posterior.reparameterize(mu, logvar, ns)
    out["samples"], out["eps_back"] = samples.numpy().copy(), eps.numpy().copy()
    out["lp_eps"] = model.posterior.log_probability(mu, logvar, eps=eps, seq_lengths=zl).numpy().copy()
    out["lp_z"] = model.posterior.log_probability(mu, logvar, z=samples, seq_lengths=zl).numpy().copy()
    out["lp_z_nolen"] = model.posterior.log_probability(mu, logvar, z=samples).numpy().copy()
    samples0, eps0 = model.posterior.reparameterize(mu, logvar, ns, tf.constant(False))
    out["samples_notrandom"] = samples0.numpy().copy()
    for flag in (False, True):
        tag = "train" if flag else "eval"
        tf.random.queue[:] = [np.asarray(eps_prior, np.float64)]
        z, lp = model.prior.sample(zl, text, tl, training=flag)
        out["prior_sample_z_" + tag], out["prior_sample_lp_" + tag] = z.numpy().copy(), lp.numpy().copy()
        tf.random.queue[:] = [np.asarray(eps_prior, np.float64)]
        z2, lp2 = model.prior(text, zl, tl, training=flag)
        out["prior_call_z_" + tag], out["prior_call_lp_" + tag] = z2.numpy().copy(), lp2.numpy().copy()
        out["prior_logprob_" + tag] = model.prior.log_probability(z, text, z_lengths=zl, condition_lengths=tl, training=flag).numpy().copy()
    tf.random.queue[:] = [np.asarray(eps_init, np.float64)]
    zi, lpi = model.prior.init(conditions=text, targets_lengths=zl, condition_lengths=tl, training=True)
    out["prior_init_z"], out["prior_init_lp"] = zi.numpy().copy(), lpi.numpy().copy()
    for k, v in leaves.items():
        if k.startswith("prior/glow/") and (k.endswith("/0/log_scale") or k.endswith("/0/bias")):
            # ActNormFlow.init assigns the variables (flow.py:194-195): read them from the model, not from the stale leaf table
            parts = k.split("/")
            out["init/" + k] = getattr(_resolve(model, parts[:-1]), parts[-1]).numpy().copy()
    return out
