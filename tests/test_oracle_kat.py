"""Known-answer tests of the oracle (CPU, no GPU).  The reference has no tests or golden vectors
(SURVEY.md section 4), so the restatement is pinned by analytic identities and by the documented
TF 2.2 semantics evaluated with plain Python loops on tiny cases."""
import math

import numpy as np
import pytest

from oracle import vaenar_numpy as O
from oracle.vaenar_numpy import Oracle
from vaenar_tts_amd.configs import LJHPS, tiny_hps
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights


def test_sequence_mask():
    assert O.sequence_mask([1, 3], 4).tolist() == [[True, False, False, False], [True, True, True, False]]
    assert O.sequence_mask([2, 3]).shape == (2, 3)          # maxlen = max(lengths) (prior.py:39-40)


def test_mask_fill_is_fp32_minus_2_pow_32():
    assert float(O.MASK_FILL) == -4294967296.0              # attention.py:240 cast to fp32


def test_positional_encoding_closed_form():
    T, D, step = 9, 8, 2.795
    pe = O.positional_encoding(T, D, step).astype(np.float64)
    for t in range(T):
        for d in range(D):
            p = t * step
            if d % 2 == 0:                                  # utils.py:351-353
                ref = math.sin(p / 10000.0 ** (d / D))
            else:                                           # utils.py:354: exponent (d-1)/D
                ref = math.cos(p / 10000.0 ** ((d - 1) / D))
            assert abs(pe[t, d] - ref) < 5e-6, (t, d)
    assert pe.dtype == np.float64 and O.positional_encoding(T, D, step).dtype == np.float32


def test_layer_norm_hand():
    x = np.array([[1.0, 2.0, 3.0, 6.0]])
    mu, var = 3.0, (4 + 1 + 0 + 9) / 4.0
    ref = (x - mu) / math.sqrt(var + 1e-3) * 2.0 + 0.5
    np.testing.assert_allclose(O.layer_norm(x, np.full(4, 2.0), np.full(4, 0.5)), ref, rtol=1e-12)


def test_batch_norm_infer_and_train():
    r = np.random.default_rng(0)
    x = r.standard_normal((2, 5, 3))
    g, b, m, v = r.standard_normal(3), r.standard_normal(3), r.standard_normal(3), r.uniform(0.5, 2, 3)
    np.testing.assert_allclose(O.batch_norm_infer(x, g, b, m, v), (x - m) / np.sqrt(v + 1e-3) * g + b, rtol=1e-12)
    y, bm, bv = O.batch_norm_train(x, g, b)
    flat = x.reshape(-1, 3)
    np.testing.assert_allclose(bm, flat.mean(0)); np.testing.assert_allclose(bv, flat.var(0))   # population var
    np.testing.assert_allclose(y, (x - bm) / np.sqrt(bv + 1e-3) * g + b, rtol=1e-12)


def test_conv1d_same_loops():
    r = np.random.default_rng(1)
    B, T, Ci, Co, k = 2, 6, 3, 4, 5
    x, w, b = r.standard_normal((B, T, Ci)), r.standard_normal((k, Ci, Co)), r.standard_normal(Co)
    ref = np.zeros((B, T, Co))
    for bb in range(B):
        for t in range(T):
            for o in range(Co):
                s = b[o]
                for j in range(k):
                    tt = t + j - k // 2                     # 2 left / 2 right zero padding
                    if 0 <= tt < T:
                        s += (x[bb, tt] * w[j, :, o]).sum()
                ref[bb, t, o] = s
    np.testing.assert_allclose(O.conv1d_same(x, w, b), ref, rtol=1e-12, atol=1e-12)
    # sequence shorter than the kernel
    np.testing.assert_allclose(O.conv1d_same(x[:, :2], w, b)[:, 0], ref2(x[:, :2], w, b)[:, 0], rtol=1e-12)


def ref2(x, w, b):
    B, T, _ = x.shape
    k = w.shape[0]
    out = np.zeros((B, T, w.shape[2]))
    for t in range(T):
        out[:, t] = b
        for j in range(k):
            tt = t + j - k // 2
            if 0 <= tt < T:
                out[:, t] += x[:, tt] @ w[j]
    return out


def _tiny():
    hps = tiny_hps()
    w = init_weights(hps, seed=7, mode="synthetic")
    return hps, w, Oracle(hps, w, np.float64)


def test_mha_masks_and_uniform_rows():
    """Key AND query AND causal masks; fully masked rows become uniform over ALL Tk keys (quirk 2)."""
    hps, w, o = _tiny()
    r = np.random.default_rng(2)
    B, T, D = 2, 6, 128
    x = r.standard_normal((B, T, D))
    lens = np.array([6, 3])
    p = "decoder/attentions/0/self_attention"
    ctx, ali = o.mha(p, x, x, lens, lens, True, 2, 1.0)
    assert np.allclose(ali.sum(-1), 1.0)
    assert np.all(ali[0][:, np.triu_indices(T, 1)[0], np.triu_indices(T, 1)[1]] == 0)      # causal
    assert np.all(ali[1, :, :3, 3:] == 0)                                                   # masked keys
    assert np.all(ali[1, :, 3:, :] == 1.0 / T)                                              # padded queries
    v = x @ o.w[p + "/value_layer/kernel"]
    np.testing.assert_allclose(ctx[1, 3:], np.broadcast_to(v[1].mean(0), (3, D)), atol=1e-12)
    # explicit loops for one (b, h, i)
    q = x @ o.w[p + "/query_layer/kernel"]; k = x @ o.w[p + "/key_layer/kernel"]
    b, h, i = 0, 1, 4
    lg = np.array([q[b, i, 64:128] @ k[b, j, 64:128] / 8.0 if j <= i else -4294967296.0 for j in range(T)])
    e = np.exp(lg - lg.max())
    np.testing.assert_allclose(ali[b, h, i], e / e.sum(), atol=1e-14)


def test_flow_roundtrip_and_logdets_cancel():
    hps, w, o = _tiny()
    b = make_batch(2, 9, 24, latent_dim=hps.Common.latent_dim, ragged=True, temperature=1.0, text_step=3, mel_step=6)
    lens = (b["mel_lengths"] + 1) // 2
    cond = np.random.default_rng(3).standard_normal((2, 9, hps.Encoder.Transformer.pre_hidden))
    z, logp = o.prior_sample(lens, cond, b["text_lengths"], b["eps"])
    # running the flow backwards recovers eps and the same log-probability (prior.py:119-152 vs :154-169)
    eps = z
    for s in reversed(range(hps.Prior.Transformer.n_blk)):
        p = f"prior/glow/{s}"
        eps, _ = o.coupling(f"{p}/2", s % 2 == 0, eps, cond, lens, b["text_lengths"], backward=True)
        eps, _ = o.invlinear_backward(f"{p}/1", eps, lens)
        eps, _ = o.actnorm_backward(f"{p}/0", eps, lens)
    np.testing.assert_allclose(eps, b["eps"], atol=1e-5)
    np.testing.assert_allclose(o.prior_log_probability(z, cond, lens, b["text_lengths"]), logp, rtol=1e-6, atol=1e-3)


def test_zero_init_coupling_is_sigmoid2_scale():
    """Reference initialisers: zero log_scale/shift heads -> scale = sigmoid(2), shift = 0 (transform.py:12-17)."""
    hps = tiny_hps()
    w = init_weights(hps, seed=1, mode="reference")
    o = Oracle(hps, w, np.float64)
    r = np.random.default_rng(4)
    z = r.standard_normal((1, 5, hps.Common.latent_dim))
    cond = r.standard_normal((1, 4, hps.Encoder.Transformer.pre_hidden))
    half = hps.Common.latent_dim // 2
    s2 = 1 / (1 + math.exp(-2.0))
    out, ld = o.coupling("prior/glow/0/2", True, z, cond, np.array([5]), np.array([4]))
    np.testing.assert_allclose(out[..., :half], z[..., :half])                   # 'upper': first half is the condition
    np.testing.assert_allclose(out[..., half:], s2 * z[..., half:], rtol=1e-12)
    np.testing.assert_allclose(ld, 5 * half * math.log(s2), rtol=1e-12)
    out, _ = o.coupling("prior/glow/1/2", False, z, cond, np.array([5]), np.array([4]))
    np.testing.assert_allclose(out[..., half:], z[..., half:])                   # 'lower': second half is the condition
    np.testing.assert_allclose(out[..., :half], s2 * z[..., :half], rtol=1e-12)


def test_actnorm_init_statistics():
    hps, w, o = _tiny()
    r = np.random.default_rng(5)
    z = 3 * r.standard_normal((2, 7, hps.Common.latent_dim)) + 1
    out, _ = o.actnorm_init("prior/glow/0/0", z, np.array([7, 4]))
    flat = out.reshape(-1, out.shape[-1])                                         # all rows, padding included
    np.testing.assert_allclose(flat.mean(0), 0, atol=1e-7)
    np.testing.assert_allclose(flat.std(0), 1, atol=1e-6)


def test_length_predictor_and_test_step_integer_arithmetic():
    hps, w, o = _tiny()
    b = make_batch(3, 11, 40, latent_dim=hps.Common.latent_dim, ragged=True, text_step=3, mel_step=7)
    mel, lens, ali = o.test_step(b["ids"], b["text_lengths"])
    pred = o.last["pred_float"]
    assert np.array_equal(lens, np.trunc(pred).astype(np.int32) + 80)            # inference.py:135,143
    assert mel.shape[1] == 2 * int(((lens + 1) // 2).max())                      # 2*ceil((pred+80)/2) frames
    # masked sum: padded characters do not contribute
    x = o.last["text_embd"]
    proj = x @ w["length_predictor/projection/kernel"].astype(np.float64) + w["length_predictor/projection/bias"]
    for i, n in enumerate(b["text_lengths"]):
        assert abs(np.exp(proj[i, :n]).sum() - pred[i]) < 1e-9


def test_posterior_head_swap_and_elbo_terms():
    """models.py:136 unpacks (mu_projection, logvar_projection) as (logvar, mu) -- quirk 1."""
    hps, w, o = _tiny()
    b = make_batch(2, 9, 20, latent_dim=hps.Common.latent_dim, ragged=True, text_step=3, mel_step=6)
    r = np.random.default_rng(6)
    mels = r.standard_normal((2, 20, hps.Audio.num_mels))
    eps = r.standard_normal((2, 1, 10, hps.Common.latent_dim))
    outs, l2, kl, ll, _ = o.call(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, False, True, eps)
    first, second = o.posterior(mels[:, ::2], o.last["text_embd"], b["text_lengths"], (b["mel_lengths"] + 1) // 2)
    np.testing.assert_allclose(o.last["logvar"], first); np.testing.assert_allclose(o.last["mu"], second)
    np.testing.assert_allclose(o.last["samples"], eps[:, 0] * np.exp(0.5 * first) + second)
    # posterior log-prob by loops (posterior.py:59-71)
    lens = (b["mel_lengths"] + 1) // 2
    for i in range(2):
        s = 0.0
        for t in range(lens[i]):
            s += -0.5 * (first.shape[2] * math.log(2 * math.pi) + (first[i, t] + eps[i, 0, t] ** 2).sum())
        assert abs(s - o.last["post_lp"][i, 0]) < 1e-8
    assert np.isfinite([l2, kl, ll]).all()
    # l2 = masked per-utterance mean of outs AND initial (models.py:184-188)
    d = ((outs - mels) ** 2).mean(-1); d0 = ((o.last["initial"] - mels) ** 2).mean(-1)
    ref = np.mean([(d[i, :n].sum() + d0[i, :n].sum()) / n for i, n in enumerate(b["mel_lengths"])])
    assert abs(ref - l2) < 1e-10


def test_decoder_reduction_factor_slicing():
    """out_projection[:, :, :rf*out_dim] then reshape [B, T*rf, out_dim] (decoder.py:193-195)."""
    hps, w, o = _tiny()
    r = np.random.default_rng(8)
    z = r.standard_normal((1, 4, hps.Common.latent_dim)); mem = r.standard_normal((1, 5, hps.Encoder.Transformer.pre_hidden))
    i2, _, _ = o.decoder(z, mem, np.array([4]), np.array([5]), 2)
    i5, _, _ = o.decoder(z, mem, np.array([4]), np.array([5]), 5)
    od = hps.Common.output_dim
    np.testing.assert_allclose(i5.reshape(1, 4, 5 * od)[:, :, :2 * od], i2.reshape(1, 4, 2 * od))


def test_padded_rectangle_leaks_into_valid_tail():
    """Convs/PostNet are unmasked (quirk 3): the last frames of a shorter utterance depend on padded rows."""
    hps, w, o = _tiny()
    b = make_batch(2, 9, 30, latent_dim=hps.Common.latent_dim, ragged=True, temperature=1.0, text_step=3, mel_step=10)
    mel, _ = o.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    eps2 = b["eps"].copy(); eps2[1, (b["mel_lengths"][1] + 1) // 2:] += 1.0     # change only padded latent frames
    mel2, _ = o.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, eps2)
    n = b["mel_lengths"][1]
    assert np.abs(mel2[1, :n - 12] - mel[1, :n - 12]).max() < 1e-12              # far from the tail: untouched
    assert np.abs(mel2[1, n - 10:n] - mel[1, n - 10:n]).max() > 1e-9             # tail leakage through 5 convs


def test_oracle_fp32_mode_tracks_fp64():
    w = init_weights(LJHPS, seed=1234, mode="synthetic", include_posterior=False)
    b = make_batch(2, 16, 40, ragged=True, temperature=1.0, text_step=5, mel_step=9)
    m64, _ = Oracle(LJHPS, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    m32, _ = Oracle(LJHPS, w, np.float32).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    assert m32.dtype == np.float32 and np.abs(m32 - m64).max() < 1e-4


def test_dropout_hash_known_answers():
    """The counter-based dropout mask shared by the engine (misc.hip: mix32 / rowop_kernel, engine.hip: site_key) and
    the oracle: fixed points of the hash, keep fraction, determinism and independence across sites."""
    from oracle.vaenar_numpy import DROPOUT_SITES, dropout_keep, dropout_site_key, _mix32
    assert int(_mix32(np.uint32(0))) == 0 and int(_mix32(np.uint32(1))) == 0x514E28B7      # murmur3 fmix32 vectors
    assert int(dropout_site_key(1234, 8)) == 3684969370
    k1 = dropout_keep((1000, 100), 0.1, 1234, DROPOUT_SITES["text_encoder/pe_dropout"])
    k2 = dropout_keep((1000, 100), 0.1, 1234, DROPOUT_SITES["text_encoder/pe_dropout"])
    k3 = dropout_keep((1000, 100), 0.1, 1234, DROPOUT_SITES["posterior/pe_dropout"])
    assert (k1 == k2).all() and abs(k1.mean() - 0.9) < 5e-3 and abs((k1 == k3).mean() - 0.82) < 1e-2
    assert dropout_keep((64,), 0.0, 1, 0).all()
    assert abs(dropout_keep((200000,), 0.5, 7, 16).mean() - 0.5) < 5e-3


def test_philox4x32_10_known_answers():
    """The counter-based generator behind vnr_random_normal (device noise, prior.py:35 / posterior.py:35): the oracle's
    restatement reproduces the published Random123 known-answer vectors of philox4x32-10."""
    from oracle.vaenar_numpy import philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in philox4x32_10([ctr], key)[0]) == want


def test_philox_normal_moments_and_streams():
    from oracle.vaenar_numpy import philox_normal
    z = philox_normal(400000, seed=1234, offset=0)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3 and abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3) < 5e-2
    # offset = a jump of whole 4-element blocks; stddev scales; a different seed is a different stream
    assert np.array_equal(philox_normal(64, 1234, offset=10), z[40:104])
    assert np.allclose(philox_normal(64, 1234, 0, stddev=0.5), 0.5 * z[:64], atol=1e-7)
    assert abs(np.corrcoef(philox_normal(100000, 1235), z[:100000])[0, 1]) < 1e-2
