"""Committed golden vectors (tests/golden/*.npz, made by oracle/make_golden.py).
CPU: the oracle still reproduces them (pins the checker against silent edits).
GPU: the HIP path reproduces them without calling the oracle at run time."""
import os

import numpy as np
import pytest

from oracle.make_golden import CASES, SEED, weights_digest
from vaenar_tts_amd.weights import init_weights

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with np.load(os.path.join(GOLD, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def _weights(name, g):
    hps = CASES[name][0]()
    w = init_weights(hps, seed=SEED, mode="synthetic")
    assert weights_digest(w) == bytes(g["weights_sha256"]).decode(), "synthetic weight generator changed"
    return hps, w


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    from oracle.vaenar_numpy import Oracle
    g = _load(name)
    hps, w = _weights(name, g)
    o = Oracle(hps, w, np.float64)
    mel, ali = o.inference(g["ids"], g["mel_lengths"], g["text_lengths"], 2, g["eps"])
    np.testing.assert_allclose(mel, g["mel"], atol=2e-6)
    np.testing.assert_allclose(o.last["text_embd"], g["text_embd"], atol=2e-6)
    np.testing.assert_allclose(o.last["z"], g["z"], atol=2e-6)
    for k, v in ali.items():
        np.testing.assert_allclose(v, g["ali_" + k], atol=1e-7)
    tmel, tlen, _ = o.test_step(g["ids"], g["text_lengths"])
    assert np.array_equal(tlen, g["ts_lengths"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_path_reproduces_golden(name):
    """north_star: mels within 1e-3 max-abs, integer frame counts bit-exact.  Asserted at 2e-4."""
    from vaenar_tts_amd.models import VAENAR
    g = _load(name)
    hps, w = _weights(name, g)
    model = VAENAR(hps, weights=w)
    try:
        eps = g["eps"] if g["eps"].any() else None
        mel, ali = model.inference(g["ids"], g["mel_lengths"], g["text_lengths"], reduction_factor=2, eps=eps,
                                   temperature=0.0 if eps is None else 1.0)
        assert np.abs(mel.numpy() - g["mel"]).max() < 2e-4
        for k in ali:
            np.testing.assert_allclose(ali[k].numpy(), g["ali_" + k], atol=1e-5)
        # test_step (inference.py:128-143)
        rf = hps.Common.final_reduction_factor
        te = model.text_encoder(g["ids"], g["text_lengths"], pos_step=np.float32(model.mel_text_len_ratio) / np.float32(rf))
        np.testing.assert_allclose(te.numpy(), g["text_embd"], atol=1e-4)
        pred = model.length_predictor(te, g["text_lengths"]).numpy()
        margin = np.abs(g["ts_pred_float"] - np.round(g["ts_pred_float"]))
        assert margin.min() > 1e-4, "fixture sits on an integer boundary"
        assert np.array_equal(pred.astype(np.int32) + 80, g["ts_lengths"])     # bit-exact frame counts
        reduced = (pred.astype(np.int32) + 80 + rf - 1) // rf
        z, _ = model.prior.sample(reduced, te, g["text_lengths"], temperature=0.0)
        _, outs, _ = model.decoder(z, te, reduced, g["text_lengths"], reduction_factor=rf)
        assert np.abs(outs.numpy() - g["ts_mel"]).max() < 2e-4
    finally:
        model.engine.close()
