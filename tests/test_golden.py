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
    """north_star: mels within 1e-3 max-abs, integer frame counts bit-exact.  Asserted at 2e-5 (round 6; 2e-4 before)."""
    from vaenar_tts_amd.models import VAENAR
    g = _load(name)
    hps, w = _weights(name, g)
    model = VAENAR(hps, weights=w)
    try:
        eps = g["eps"] if g["eps"].any() else None
        mel, ali = model.inference(g["ids"], g["mel_lengths"], g["text_lengths"], reduction_factor=2, eps=eps,
                                   temperature=0.0 if eps is None else 1.0)
        assert np.abs(mel.numpy() - g["mel"]).max() < 2e-5
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
        assert np.abs(outs.numpy() - g["ts_mel"]).max() < 2e-5
    finally:
        model.engine.close()


# ---- fixtures produced by the reference's own Python over oracle/tf_shim (composition pin) ---------------------------
from oracle.make_golden import REF_CASES


def _ref_weights(name, g):
    mk, _, seed = REF_CASES[name]
    hps = mk()
    w = init_weights(hps, seed=seed, mode="synthetic")
    assert weights_digest(w) == bytes(g["weights_sha256"]).decode()
    return hps, w


@pytest.mark.parametrize("name", sorted(REF_CASES))
def test_oracle_matches_reference_python(name):
    """models.VAENAR.inference / .call of the reference (run over the tf shim, float64) vs the oracle."""
    from oracle.vaenar_numpy import Oracle
    g = _load(name)
    hps, w = _ref_weights(name, g)
    o = Oracle(hps, w, np.float64)
    mel, ali = o.inference(g["ids"], g["mel_lengths"], g["text_lengths"], 2, g["eps"])
    # the oracle stages the positional encoding in float32 like TF; the shim run is float64 -> ~1e-6 apart
    np.testing.assert_allclose(mel, g["mel"], atol=5e-6)
    for k, v in ali.items():
        np.testing.assert_allclose(v, g["ali_" + k], atol=1e-6)
    outs, l2, kl, ll, _ = o.call(g["ids"], g["call_mels"], g["mel_lengths"], g["text_lengths"], 2, False, False,
                                 g["call_eps"].astype(np.float64))
    np.testing.assert_allclose(outs, g["call_outs"], atol=5e-6)
    np.testing.assert_allclose(l2, g["call_l2"], rtol=1e-5)
    np.testing.assert_allclose(ll, g["call_length"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(kl, g["call_kl"], rtol=1e-5, atol=2e-2)   # float32-cast slogdet in the oracle (flow.py:127)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference checkout only exists in the build container")
def test_reference_python_live_against_oracle():
    """Re-runs the reference's modules live (this container only) on a fresh seed."""
    from oracle.run_reference_on_shim import reference_inference
    from oracle.vaenar_numpy import Oracle
    from vaenar_tts_amd.configs import tiny_hps
    from vaenar_tts_amd.synthetic import make_batch
    hps = tiny_hps()
    w = init_weights(hps, seed=99, mode="synthetic")
    b = make_batch(2, 9, 26, latent_dim=hps.Common.latent_dim, ragged=True, temperature=1.0, seed=99, text_step=4, mel_step=9)
    mel, ali, _, _ = reference_inference(hps, w, b["ids"], b["mel_lengths"], b["text_lengths"], b["eps"])
    rm, ra = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    np.testing.assert_allclose(mel, rm, atol=5e-6)
    for k in ra:
        np.testing.assert_allclose(ali[k], ra[k], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(REF_CASES))
def test_hip_path_matches_reference_python(name):
    """The HIP path against vectors produced by the reference's own Python (inference mels + ELBO terms)."""
    from vaenar_tts_amd.models import VAENAR
    g = _load(name)
    hps, w = _ref_weights(name, g)
    model = VAENAR(hps, weights=w)
    try:
        mel, ali = model.inference(g["ids"], g["mel_lengths"], g["text_lengths"], reduction_factor=2, eps=g["eps"])
        assert np.abs(mel.numpy() - g["mel"]).max() < 2e-5
        for k in ali:
            np.testing.assert_allclose(ali[k].numpy(), g["ali_" + k], atol=1e-5)
        outs, l2, kl, ll, _ = model(g["ids"], g["call_mels"], g["mel_lengths"], g["text_lengths"], reduction_factor=2,
                                    training=False, reduce_loss=False, eps=g["call_eps"])
        assert np.abs(outs.numpy() - g["call_outs"]).max() < 2e-5
        np.testing.assert_allclose(l2.numpy(), g["call_l2"], rtol=1e-4)
        np.testing.assert_allclose(ll.numpy(), g["call_length"], rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(kl.numpy(), g["call_kl"], rtol=1e-3, atol=6e-2)
    finally:
        model.engine.close()


# ---- training step fixture (tests/golden/train_tiny.npz, made by oracle/make_golden.py:build_train) ----------------------
def _train_fixture():
    from vaenar_tts_amd.configs import tiny_hps
    g = _load("train_tiny")
    hps = tiny_hps()
    w = init_weights(hps, seed=SEED, mode="synthetic")
    assert weights_digest(w) == bytes(g["weights_sha256"]).decode(), "synthetic weight generator changed"
    return g, hps, w


def test_autograd_oracle_reproduces_training_golden():
    from oracle.make_golden import digest
    from oracle.vaenar_torch import TorchOracle
    g, hps, w = _train_fixture()
    grads, sc = TorchOracle(hps, w).gradients(g["ids"], g["mels"], g["mel_lengths"], g["text_lengths"], 2, g["eps"],
                                              kl_weight=float(g["kl_weight"]), length_weight=hps.Train.length_weight,
                                              dropout_seed=int(g["dropout_seed"]))
    np.testing.assert_allclose([sc["loss"], sc["mel_l2"], sc["kl"], sc["length_l2"]], g["scalars"], rtol=1e-10)
    for k in grads:
        d, ref = digest(grads[k]), g["grad/" + k]
        assert np.abs(d - ref).max() <= 1e-8 * max(ref[18], 1e-30) * max(1.0, np.sqrt(grads[k].size)) + 1e-13, k


@pytest.mark.gpu
def test_gpu_training_step_reproduces_golden():
    """The HIP training step against the committed fixture -- no oracle at run time."""
    from oracle.make_golden import digest
    from vaenar_tts_amd.models import VAENAR
    g, hps, w = _train_fixture()
    model = VAENAR(hps, weights=w)
    try:
        out = model.train_step(g["ids"], g["mels"], g["text_lengths"], g["mel_lengths"], float(g["kl_weight"]), 2, eps=g["eps"],
                               dropout_seed=int(g["dropout_seed"]), apply_update=False)
        grads = model.gradients()
    finally:
        model.engine.close()
    np.testing.assert_allclose(out, g["scalars"], rtol=1e-4)
    for k in grads:
        d, ref = digest(grads[k]), g["grad/" + k]
        mx = max(ref[18], 1e-30)
        assert np.abs(d[:16] - ref[:16]).max() <= 2e-3 * mx + 1e-7, k                    # sampled entries
        assert abs(d[17] - ref[17]) <= 2e-3 * ref[17] + 1e-7 and abs(d[18] - ref[18]) <= 2e-3 * mx + 1e-7, k   # l2 norm, max


def test_n_sample_2_fixture_equals_tiled_single_sample_oracle():
    """tests/golden/refshim_nsample2.npz (the reference's own Python with hps.Train.num_samples = 2) against the NumPy oracle run
    on the batch tiled n_sample times: the equivalence the engine-side implementation of n_sample > 1 relies on."""
    from oracle.vaenar_numpy import Oracle
    from vaenar_tts_amd.configs import tiny_hps
    g = _load("refshim_nsample2")
    hps = tiny_hps()
    w = init_weights(hps, seed=7, mode="synthetic")
    assert weights_digest(w) == bytes(g["weights_sha256"]).decode()
    n = int(g["n_sample"])
    rep = lambda a: np.repeat(a, n, axis=0)      # noqa: E731
    B, _, Tz, C = g["eps"].shape
    outs, l2, kl, ll, _ = Oracle(hps, w, np.float64).call(rep(g["ids"]), rep(g["mels"]), rep(g["mel_lengths"]), rep(g["text_lengths"]), 2,
                                                          False, False, g["eps"].reshape(B * n, 1, Tz, C).astype(np.float64))
    np.testing.assert_allclose(outs, g["outs"], atol=5e-6)
    np.testing.assert_allclose(l2.reshape(B, n).mean(1), g["l2"], rtol=1e-5)
    np.testing.assert_allclose(ll.reshape(B, n)[:, 0], g["length"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(kl.reshape(B, n).mean(1), g["kl"], rtol=1e-5, atol=2e-2)
