"""The arithmetic contract of the split-fp16 path (include/vaenar_hip.h, "Arithmetic contract of the split path").

TensorFlow evaluates every Dense / Conv1D / matmul of the path in fp32 (/root/reference/modules/attention.py:217-246,
modules/utils.py:44-53,76-85): activations of 1e5 or 1e-5 are ordinary there.  The engine's default path splits activations into
UNSCALED fp16 hi/lo pairs, which keeps fp32-class accuracy only inside a magnitude window.  These tests pin
  (i)   the exact-fp32 mode (`split_fp16 = 0`) at model level against the float64 oracle and the reference-on-shim fixtures;
  (ii)  the range guard: a model whose weights are rescaled so that a GEMM input tensor reaches 1e5 or 1e-5 -- mathematically the
        same function, which the oracle and the reference evaluate without trouble -- must still match (the engine surveys the
        ranges on the first call and keeps those modules on exact fp32 MFMA), an in-window model must stay on the split path,
        and an out-of-window attention operand must be REFUSED with a message, never answered with an inf.
(The operator-level criterion relative to the row's own scale lives in tests/test_gpu_ops.py::test_dense_split_fp16.)"""
import numpy as np
import pytest

from oracle.vaenar_numpy import Oracle
from vaenar_tts_amd import _lib
from vaenar_tts_amd.configs import LJHPS, tiny_hps
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

pytestmark = pytest.mark.gpu
MEL_TOL = 2e-4          # asserted; the contract is 1e-3 (north star)


def _hps(name):
    return tiny_hps() if name == "tiny" else LJHPS


def _batch(hps, name):
    if name == "tiny":
        return make_batch(3, 11, 40, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True,
                          temperature=1.0, text_step=3, mel_step=7)
    return make_batch(4, 37, 150, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True,
                      temperature=1.0, text_step=5, mel_step=23)


def _gemm_launches(model, fn):
    e = model.engine
    e.profile(True); e.profile_reset()
    try:
        out = fn()
        e.synchronize()
        return out, e.profile_get("gemm")["launches"] + e.profile_get("chain")["launches"] + e.profile_get("chain_ali")["launches"], \
            e.profile_get("gemm_fp32")["launches"]
    finally:
        e.profile(False)


@pytest.mark.parametrize("name", ["tiny", "lj"])
def test_exact_fp32_mode_matches_the_oracle(name):
    """`split_fp16 = 0` -- every Dense / Conv1D product on v_mfma_f32_32x32x2_f32, the mode bench.py reports as `exact_fp32` and the
    one the range guard falls back to -- at model level: inference, alignments, frame counts and the ELBO forward."""
    hps = _hps(name)
    w = init_weights(hps, seed=1234, mode="synthetic")
    model, oracle = VAENAR(hps, weights=w), Oracle(hps, w, np.float64)
    try:
        model.engine.set_option("split_fp16", 0)
        b = _batch(hps, name)
        (mel, ali), n_split, n_exact = _gemm_launches(model, lambda: model.inference(b["ids"], b["mel_lengths"], b["text_lengths"],
                                                                                     reduction_factor=2, eps=b["eps"]))
        assert n_split == 0 and n_exact > 0, (n_split, n_exact)
        rmel, rali = oracle.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
        err = np.abs(mel.numpy() - rmel).max()
        print(f"[{name}] exact fp32: max-abs mel err {err:.3e}")
        assert err < MEL_TOL
        for k in rali:
            np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)
        # text encoder -> length predictor -> integer frame counts (inference.py:135)
        pos_step = np.float32(hps.Common.mel_text_len_ratio) / np.float32(2)
        te = model.text_encoder(b["ids"], b["text_lengths"], pos_step=pos_step)
        pl = model.length_predictor(te, b["text_lengths"]).numpy()
        rl = oracle.length_predictor(oracle.text_encoder(b["ids"], b["text_lengths"], pos_step=pos_step), b["text_lengths"])
        np.testing.assert_allclose(pl, rl, rtol=2e-5)
        # ELBO forward (models.py:105-197)
        r = np.random.Generator(np.random.PCG64(5))
        mels = r.standard_normal((len(b["mel_lengths"]), int(b["mel_lengths"].max()), hps.Audio.num_mels)).astype(np.float32)
        Tz = (int(b["mel_lengths"].max()) + 1) // 2
        eps = r.standard_normal((len(b["mel_lengths"]), 1, Tz, hps.Common.latent_dim)).astype(np.float32)
        outs, l2, kl, ll, _ = model(b["ids"], mels, b["mel_lengths"], b["text_lengths"], reduction_factor=2, training=False,
                                    reduce_loss=False, eps=eps)
        ro, rl2, rkl, rll, _ = oracle.call(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, False, False, eps.astype(np.float64))
        assert np.abs(outs.numpy() - ro).max() < MEL_TOL
        np.testing.assert_allclose(l2.numpy(), rl2, rtol=1e-4)
        np.testing.assert_allclose(kl.numpy(), rkl, rtol=1e-3, atol=6e-2)
    finally:
        model.engine.close()


def test_exact_fp32_mode_matches_the_reference_fixture():
    """The same mode against vectors the reference's own Python produced (tests/golden/refshim_lj.npz)."""
    from test_golden import _load, _ref_weights
    g = _load("refshim_lj")
    hps, w = _ref_weights("refshim_lj", g)
    model = VAENAR(hps, weights=w)
    try:
        model.engine.set_option("split_fp16", 0)
        mel, ali = model.inference(g["ids"], g["mel_lengths"], g["text_lengths"], reduction_factor=2, eps=g["eps"])
        assert np.abs(mel.numpy() - g["mel"]).max() < MEL_TOL
        for k in ali:
            np.testing.assert_allclose(ali[k].numpy(), g["ali_" + k], atol=1e-5)
    finally:
        model.engine.close()


def _rescaled(w, pairs):
    """A copy of the weights with (path, factor) products applied -- the caller picks pairs that leave the function unchanged."""
    out = dict(w)
    for path, f in pairs:
        assert path in out, path
        out[path] = (np.asarray(out[path], np.float64) * f).astype(np.float32)
    return out


FFN0 = "prior/glow/0/2/net/attentions/0/ffn/"
DEC_FFN = "decoder/attentions/1/ffn/"


@pytest.mark.parametrize("factor", [2.0 ** 17, 2.0 ** -17], ids=["hidden-1e5", "hidden-1e-5"])
def test_range_guard_keeps_a_rescaled_model_exact(factor):
    """FFN of a prior block and of a decoder block (utils.py:44-53): dense1 (kernel and bias) x f, dense2 kernel / f.  ReLU is positively
    homogeneous, so the network computes the SAME function -- but its hidden activations, the GEMM input of dense2, are now of order
    1e5 (fp16 overflows at 65504) or 1e-5 (below fp16's normal range).  The float64 oracle, like TensorFlow's fp32, does not care.  The
    engine must notice on its first call (range survey), keep the prior and the decoder on exact fp32 MFMA, and match the oracle of the
    RESCALED weights; powers of two keep the rescaled function bit-identical in exact arithmetic."""
    hps = LJHPS
    w0 = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    pairs = []
    for p in (FFN0, DEC_FFN):
        pairs += [(p + "dense1/kernel", factor), (p + "dense1/bias", factor), (p + "dense2/kernel", 1.0 / factor)]
    w = _rescaled(w0, pairs)
    b = _batch(hps, "lj")
    ref0, _ = Oracle(hps, w0, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    ref, rali = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    assert np.abs(ref - ref0).max() < 1e-9                       # the rescaling is neutral
    model = VAENAR(hps, weights=w)
    try:
        mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        got = mel.numpy()
        assert np.isfinite(got).all()
        info = model.engine.range_info()
        print("range guard:", info)
        assert info["encoder"] == 2 and info["prior"] == 2 and info["decoder"] == 2 and info["surveys"] == 1
        assert (info["hi"] >= 32768.0) if factor > 1 else (0 < info["lo"] < 2.0 ** -6)
        assert np.abs(got - ref).max() < MEL_TOL
        for k in rali:
            np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)
        # later calls stay on the exact path without another survey
        (mel2, _), n_split, n_exact = _gemm_launches(model, lambda: model.inference(b["ids"], b["mel_lengths"], b["text_lengths"],
                                                                                    reduction_factor=2, eps=b["eps"]))
        assert n_split == 0 and n_exact > 0 and model.engine.range_info()["surveys"] == 1
        assert np.array_equal(mel2.numpy(), got)
        # with the guard off the same weights show what it protects against: inf / NaN (1e5) or a mel error far outside fp32 class (1e-5)
        model.engine.set_option("range_guard", 0)
        raw, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        bad = raw.numpy()
        unguarded = np.inf if not np.isfinite(bad).all() else np.abs(bad - ref).max()
        print(f"unguarded split path on the rescaled weights: max-abs mel err {unguarded:.3e}")
        assert unguarded > 20 * np.abs(got - ref).max()
    finally:
        model.engine.close()


def test_range_guard_leaves_an_in_window_model_on_the_split_path():
    """The synthetic LJ weights are inside the window: one survey, then the split path (chains, split GEMMs), bit-identical to a run with
    the guard switched off; uploading a weight afterwards triggers exactly one more survey."""
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    b = _batch(hps, "lj")
    model = VAENAR(hps, weights=w)
    try:
        mel, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        info = model.engine.range_info()
        print("range guard:", info)
        assert (info["encoder"], info["prior"], info["decoder"], info["surveys"]) == (1, 1, 1, 1)
        assert 2.0 ** -6 <= info["lo"] and info["hi"] < 32768.0
        (mel2, _), n_split, n_exact = _gemm_launches(model, lambda: model.inference(b["ids"], b["mel_lengths"], b["text_lengths"],
                                                                                    reduction_factor=2, eps=b["eps"]))
        assert n_split > 0 and n_exact == 0 and model.engine.range_info()["surveys"] == 1
        assert np.array_equal(mel.numpy(), mel2.numpy())         # the surveying call returned the split path's result
        model.engine.set_option("range_guard", 0)
        mel3, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        assert np.array_equal(mel.numpy(), mel3.numpy())
        model.engine.set_option("range_guard", 1)
        k = "decoder/residual_projection/bias"
        model.engine.set_weight(k, np.asarray(w[k], np.float32))
        model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        assert model.engine.range_info()["surveys"] == 2
    finally:
        model.engine.close()


def test_range_guard_refuses_an_out_of_window_attention_operand():
    """Query kernel x 2^18, key kernel / 2^18 in a decoder block: the logits are unchanged, but the attention core would split a query of
    order 1e5 -- it has no exact-fp32 form, so the call must fail with a message that names the range (never an inf)."""
    hps = LJHPS
    w0 = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    p = "decoder/attentions/0/cross_attention/"
    w = _rescaled(w0, [(p + "query_layer/kernel", 2.0 ** 18), (p + "key_layer/kernel", 2.0 ** -18)])
    b = _batch(hps, "lj")
    model = VAENAR(hps, weights=w)
    try:
        with pytest.raises(_lib.VnrError) as ei:
            model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        assert "attention operand" in str(ei.value) and "window" in str(ei.value)
    finally:
        model.engine.close()
