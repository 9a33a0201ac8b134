#!/usr/bin/env python3
"""Drive the REFERENCE's own Python (models.VAENAR + modules.*) over oracle/tf_shim and return its outputs.
This container only (/root/reference is not present on the GPU box)."""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_reference():
    """Import the reference packages with `tensorflow` resolved to the shim."""
    for m in [k for k in sys.modules if k == "tensorflow" or k.split(".")[0] in ("modules", "models", "configs")]:
        del sys.modules[m]
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(HERE, "tf_shim"))
    try:
        tf = importlib.import_module("tensorflow")
        assert tf.__file__.startswith(HERE), "a real tensorflow is installed; the shim is not needed"
        models = importlib.import_module("models.models")
        hp = importlib.import_module("configs.hparams")
    finally:
        sys.path.remove(REF)
        sys.path.remove(os.path.join(HERE, "tf_shim"))
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
