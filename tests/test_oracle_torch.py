"""The autograd restatement (oracle/vaenar_torch.py) is pinned against the NumPy specification: identical forward on
the training-mode ELBO (dropout ON, BN batch statistics), and its gradients against central finite differences of the
NumPy oracle's loss for a handful of variables."""
import numpy as np
import pytest

from oracle.vaenar_numpy import Oracle
from oracle.vaenar_torch import TorchOracle, adam_step
from vaenar_tts_amd.configs import tiny_hps
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights


def _case():
    hps = tiny_hps()
    w = init_weights(hps, seed=1234, mode="synthetic")
    b = make_batch(2, 7, 18, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                   ragged=True, text_step=2, mel_step=5)
    r = np.random.Generator(np.random.PCG64(3))
    Tm = int(b["mel_lengths"].max())
    mels = r.standard_normal((2, Tm, hps.Audio.num_mels))
    eps = r.standard_normal((2, 1, (Tm + 1) // 2, hps.Common.latent_dim))
    return hps, w, b, mels, eps


def _np_loss(hps, w, b, mels, eps, seed=9, kl_weight=1e-5, length_weight=1.0):
    o = Oracle(hps, {k: np.asarray(v, np.float64) for k, v in w.items()}, np.float64)
    o.dropout_seed = seed
    _, l2, kl, ll, _ = o.call(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, True, True, eps)
    return float(l2 + kl_weight * max(kl, 0.0) + length_weight * ll), (float(l2), float(kl), float(ll))


@pytest.mark.parametrize("inverse", [False, True])
def test_torch_forward_equals_numpy_oracle(inverse):
    """(inverse: Prior.Transformer.inverse = True -- BaseFlow.bwd_pass then runs the _forward passes, /root/reference/modules/flow.py:91-113)"""
    hps, w, b, mels, eps = _case()
    hps.Prior.Transformer.inverse = inverse
    t = TorchOracle(hps, w)
    loss, mel_l2, kl, ll = t.train_loss(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, dropout_seed=9)
    ref, (rl2, rkl, rll) = _np_loss(hps, w, b, mels, eps)
    assert abs(float(mel_l2) - rl2) < 1e-10 and abs(float(ll) - rll) < 1e-10
    assert abs(float(kl) - rkl) < 1e-7 * max(1.0, abs(rkl))
    assert abs(float(loss) - ref) < 1e-9


@pytest.mark.parametrize("path", ["decoder/residual_projection/bias", "posterior/mu_projection/kernel",
                                  "prior/glow/0/1/weight", "prior/glow/1/0/log_scale",
                                  "text_encoder/prenet/conv_stack/0/bn/gamma", "length_predictor/projection/kernel",
                                  "decoder/attentions/0/cross_attention/key_layer/kernel"])
def test_torch_gradient_matches_finite_difference(path):
    hps, w, b, mels, eps = _case()
    kw = 1.0 if path.startswith("prior") else 1e-5         # make the KL term visible for the flow variables
    # the length predictor sees stop_gradient(text_embd) (models.py:133): a finite difference of the encoder variables
    # would see through it, so the length term is switched off for them
    lw = 0.0 if path.startswith("text_encoder") else 1.0
    g, _ = TorchOracle(hps, w).gradients(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, kl_weight=kw,
                                         length_weight=lw, dropout_seed=9)
    idx = np.unravel_index(int(np.abs(g[path]).argmax()), g[path].shape) if g[path].ndim else ()
    h = 1e-5
    vals = []
    for sgn in (+1, -1):
        w2 = {k: np.asarray(v, np.float64).copy() for k, v in w.items()}
        w2[path][idx] += sgn * h
        vals.append(_np_loss(hps, w2, b, mels, eps, kl_weight=kw, length_weight=lw)[0])
    fd = (vals[0] - vals[1]) / (2 * h)
    assert abs(fd - g[path][idx]) <= 2e-5 * max(1.0, abs(fd)), (fd, g[path][idx])


def test_adam_step_known_answer():
    w = {"a": np.array([1.0, -2.0])}; g = {"a": np.array([0.5, -0.25])}
    m = {"a": np.zeros(2)}; v = {"a": np.zeros(2)}
    adam_step(w, g, m, v, 1, lr=1e-3)
    # first step of bias-corrected Adam moves every coordinate by lr * sign(g) (up to eps)
    np.testing.assert_allclose(w["a"], [1.0 - 1e-3, -2.0 + 1e-3], atol=1e-8)


@pytest.mark.parametrize("inverse", [False, True])
def test_torch_inference_equals_numpy_oracle(inverse):
    """The torch restatement's inference path (the timed CPU baseline of bench.py) against the NumPy specification."""
    hps, w, b, mels, eps = _case()
    hps.Prior.Transformer.inverse = inverse
    z_eps = np.random.Generator(np.random.PCG64(4)).standard_normal((2, (int(b["mel_lengths"].max()) + 1) // 2, hps.Common.latent_dim))
    ref, _ = Oracle(hps, {k: np.asarray(v, np.float64) for k, v in w.items()}, np.float64).inference(
        b["ids"], b["mel_lengths"], b["text_lengths"], 2, z_eps)
    got = TorchOracle(hps, w).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, z_eps).numpy()
    np.testing.assert_allclose(got, ref, atol=1e-10)
    import torch
    got32 = TorchOracle(hps, w, torch.float32).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, z_eps).numpy()
    np.testing.assert_allclose(got32, ref, atol=2e-4)
    TorchOracle(hps, w)          # back to the float64 working dtype for the other tests


def test_kink_aware_comparison_accepts_a_flipped_unit_and_rejects_a_defect():
    """oracle/kinks.py on the CPU: an "implementation" whose gradient is the oracle's with the ONE hidden unit closest to its ReLU kink on
    the other side (+ rounding noise) passes and the unit is named; the plain criterion alone would have failed it; a gradient with one
    tensor off by 1 % is rejected, with and without such a unit."""
    from oracle import kinks
    hps, w, b, mels, eps = _case()
    run = kinks.torch_oracle_run(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, 1.0, 11)
    g0, sc, pre = run(None)
    margin, site, idx = min((float(np.abs(v).min()), k, int(np.abs(v).reshape(-1).argmin())) for k, v in pre.items() if k.endswith("ffn/dense1/bias"))
    g1 = run({site: [idx]})[0]
    r = np.random.Generator(np.random.PCG64(1))
    impl = {k: v + 1e-6 * np.abs(v).max() * r.standard_normal(v.shape) for k, v in g1.items()}
    assert kinks._bad(impl, g0, 2e-3, 1e-7), "the flip of the closest unit is invisible at 2e-3: choose another case"
    sc2, flipped = kinks.compare(impl, run, tau=2 * margin + 1e-12)
    assert sc2 == sc and [(s, i) for s, i, _ in flipped] == [(site, idx)]
    clean = {k: v + 1e-6 * np.abs(v).max() * r.standard_normal(v.shape) for k, v in g0.items()}
    assert kinks.compare(clean, run) == (sc, [])
    victim = "posterior/attentions/0/att_proj2/kernel"
    for base in (clean, impl):
        broken = dict(base); broken[victim] = base[victim] * 1.01
        with pytest.raises(AssertionError):
            kinks.compare(broken, run, tau=2 * margin + 1e-12)
