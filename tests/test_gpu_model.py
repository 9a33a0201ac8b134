"""Module- and model-level parity of the HIP path (through the Python mirror of the reference call
surface -> ctypes -> C ABI) against the float64 NumPy oracle on the same seeded inputs.

north_star tolerance: mels within 1e-3 max-abs (fp32), integer frame counts bit-exact.
Observed differences are ~1e-5; the asserts use 2e-4 so that a real regression is caught long
before the contractual 1e-3."""
import numpy as np
import pytest

from oracle.vaenar_numpy import Oracle
from vaenar_tts_amd.configs import LJHPS, tiny_hps
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

pytestmark = pytest.mark.gpu

MEL_TOL = 2e-5          # asserted (round 6: was 2e-4 with 3e-6 .. 6e-6 measured -- 40x of slack hid a 1.3e-4 defect in round 5); contract is 1e-3
CONTRACT_TOL = 1e-3


def _setup(name):
    hps = tiny_hps() if name == "tiny" else LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic")
    model = VAENAR(hps, weights=w)
    oracle = Oracle(hps, w, np.float64)
    return hps, model, oracle


@pytest.fixture(scope="module", params=["tiny", "lj"])
def setup(request):
    hps, model, oracle = _setup(request.param)
    yield request.param, hps, model, oracle
    model.engine.close()


def _batch(hps, name, temperature=1.0):
    if name == "tiny":
        return make_batch(3, 11, 40, vocab_size=hps.Encoder.Transformer.vocab_size,
                          latent_dim=hps.Common.latent_dim, ragged=True, temperature=temperature,
                          text_step=3, mel_step=7)
    return make_batch(4, 37, 150, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                      ragged=True, temperature=temperature, text_step=5, mel_step=23)


def test_text_encoder_and_length_predictor(setup):
    name, hps, model, oracle = setup
    b = _batch(hps, name)
    pos_step = np.float32(hps.Common.mel_text_len_ratio) / np.float32(2)
    got = model.text_encoder(b["ids"], b["text_lengths"], pos_step=pos_step, training=False)
    ref = oracle.text_encoder(b["ids"], b["text_lengths"], pos_step=pos_step)
    np.testing.assert_allclose(got.numpy(), ref, atol=1e-4, rtol=0)
    # length predictor on the GPU's own encoding, compared with the oracle on the oracle's encoding
    pl = model.length_predictor(got, b["text_lengths"]).numpy()
    rl = oracle.length_predictor(ref, b["text_lengths"])
    np.testing.assert_allclose(pl, rl, rtol=2e-5)
    # integer frame counts (inference.py:135): bit-exact whenever the oracle is not within the fp32
    # noise band of an integer boundary
    margin = np.minimum(rl - np.floor(rl), np.ceil(rl) - rl)
    safe = margin > 1e-3 * np.maximum(1.0, rl) * 1e-1
    assert np.array_equal(pl.astype(np.int32)[safe], rl.astype(np.float32).astype(np.int32)[safe])
    assert safe.sum() >= 1


def test_prior_sample(setup):
    name, hps, model, oracle = setup
    b = _batch(hps, name)
    rf = 2
    reduced = (b["mel_lengths"] + rf - 1) // rf
    pos_step = np.float32(hps.Common.mel_text_len_ratio) / np.float32(rf)
    text_embd = oracle.text_encoder(b["ids"], b["text_lengths"], pos_step=pos_step)
    z, logp = model.prior.sample(reduced, text_embd.astype(np.float32), b["text_lengths"], training=False,
                                 eps=b["eps"])
    rz, rlogp = oracle.prior_sample(reduced, text_embd.astype(np.float32).astype(np.float64), b["text_lengths"],
                                    b["eps"])
    np.testing.assert_allclose(z.numpy(), rz, atol=1e-4, rtol=0)
    np.testing.assert_allclose(logp.numpy(), rlogp, rtol=2e-5, atol=1e-2)
    # temperature 0 (inference.py:95 default): eps = None means exact zeros
    z0, _ = model.prior.sample(reduced, text_embd.astype(np.float32), b["text_lengths"], temperature=0.0)
    rz0, _ = oracle.prior_sample(reduced, text_embd.astype(np.float32).astype(np.float64), b["text_lengths"],
                                 np.zeros_like(b["eps"]))
    np.testing.assert_allclose(z0.numpy(), rz0, atol=1e-4, rtol=0)


@pytest.mark.parametrize("rf", [2, 5])
def test_decoder(setup, rf):
    name, hps, model, oracle = setup
    b = _batch(hps, name)
    r = np.random.Generator(np.random.PCG64(7))
    reduced = (b["mel_lengths"] + rf - 1) // rf
    Tz = int(reduced.max())
    z = r.standard_normal((len(reduced), Tz, hps.Common.latent_dim)).astype(np.float32)
    mem = r.standard_normal((len(reduced), b["ids"].shape[1], hps.Encoder.Transformer.pre_hidden)).astype(np.float32)
    ini, out, ali = model.decoder(z, mem, reduced, b["text_lengths"], reduction_factor=rf, training=False)
    rini, rout, rali = oracle.decoder(z.astype(np.float64), mem.astype(np.float64), reduced, b["text_lengths"], rf)
    np.testing.assert_allclose(ini.numpy(), rini, atol=MEL_TOL, rtol=0)
    np.testing.assert_allclose(out.numpy(), rout, atol=MEL_TOL, rtol=0)
    assert sorted(ali.keys()) == sorted(rali.keys())
    for k in rali:
        np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("temperature", [0.0, 1.0])
def test_inference_matches_oracle(setup, fused, temperature):
    """VAENAR.inference (models.py:199-210) end to end on a ragged batch; padded rectangle included."""
    name, hps, model, oracle = setup
    b = _batch(hps, name, temperature)
    eps = b["eps"] if temperature else None
    mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=eps,
                               temperature=temperature, fused=fused)
    rmel, rali = oracle.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    got = mel.numpy()
    err_all = np.abs(got - rmel).max()
    valid = np.arange(got.shape[1])[None, :] < b["mel_lengths"][:, None]
    err_valid = np.abs(got - rmel)[valid].max()
    print(f"[{name}] fused={fused} T={temperature} max-abs mel err: valid {err_valid:.3e} rectangle {err_all:.3e}")
    assert err_all < MEL_TOL < CONTRACT_TOL
    for k in rali:
        np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)


@pytest.mark.parametrize("opts", [{}, {"attn_presplit": 0}, {"attn_presplit_self": 0}, {"late_dec_kv": 0}, {"chain": 0},
                                  {"chain_rows64": 1, "gemm_wide_tiles": 1}],
                         ids=["default", "cross-fp32", "self-fp32", "kv-early", "no-chain", "throughput-tiles"])
def test_inference_long_text_and_ab_switches(opts):
    """T_text = 150 > 128: the cross-attention leaves the operand-image kernel (attention3, Tk <= 128) for the fp32-operand
    kernel (attention2: stored probabilities up to 512 keys, the two-pass form beyond); T_z = 45 is not a multiple of 16, so the V stage
    of the chain tails takes the generic image scatter.  Every A/B switch of the attention path gives the same mels."""
    hps, model, oracle = _setup("tiny")
    try:
        for k, v in opts.items():
            model.engine.set_option(k, v)
        b = make_batch(2, 150, 90, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                       ragged=True, temperature=1.0, text_step=21, mel_step=17)
        mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        rmel, rali = oracle.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
        assert np.abs(mel.numpy() - rmel).max() < MEL_TOL
        for k in rali:
            np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)
        # ... and a short text (images everywhere) on the same handle
        b = make_batch(3, 40, 64, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                       ragged=True, temperature=1.0, text_step=7, mel_step=9)
        mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        rmel, rali = oracle.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
        assert np.abs(mel.numpy() - rmel).max() < MEL_TOL
        for k in rali:
            np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)
    finally:
        model.engine.close()


def test_inference_beyond_512_latent_frames_and_200_tokens():
    """Sizes past every tile limit of the fast paths (SURVEY 8c "maximum sizes"): T_text = 200 (two key blocks more than the
    operand-image cross-attention takes), T_mel = 1200 -> T_z = 600 latent frames (causal self-attention over more than 512 keys,
    19 row panels per utterance, the last one partial), ragged lengths, alignments returned [B, 4, 600, 200]."""
    hps, model, oracle = _setup("tiny")
    try:
        b = make_batch(2, 200, 1200, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                       ragged=True, temperature=1.0, text_step=37, mel_step=171)
        mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        rmel, rali = oracle.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
        assert mel.shape == (2, 1200, hps.Audio.num_mels)
        assert np.abs(mel.numpy() - rmel).max() < MEL_TOL
        for k in rali:
            np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)
    finally:
        model.engine.close()


@pytest.mark.parametrize("B,Tt,Tm,rows64", [
    (1, 32, 64, 0), (2, 33, 62, 1), (3, 127, 66, 0), (2, 128, 128, 1), (2, 129, 130, 0), (5, 5, 32, 1), (1, 64, 96, 1), (4, 31, 34, 0),
])
def test_tile_boundaries_of_the_operand_images(B, Tt, Tm, rows64):
    """Lengths that sit on, just below and just above the 32-row image tiles, the 16-row half tiles of the un-transposed V stage
    and the Tk <= 128 limit of the cross-attention image kernel; ragged and full; both chain panel heights."""
    hps, model, oracle = _setup("tiny")
    try:
        model.engine.set_option("chain_rows64", rows64)
        model.engine.set_option("gemm_wide_tiles", rows64)
        for ragged in (False, True):
            b = make_batch(B, Tt, Tm, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                           ragged=ragged, temperature=1.0, text_step=max(1, Tt // 7), mel_step=max(2, Tm // 9), seed=Tt * 1000 + Tm)
            mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
            rmel, rali = oracle.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
            assert np.abs(mel.numpy() - rmel).max() < MEL_TOL, (B, Tt, Tm, ragged)
            for k in rali:
                np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)
    finally:
        model.engine.close()


def test_test_step_frame_counts(setup):
    """inference.py:128-143: length predictor -> int32 trunc -> +80 -> ceil(/2) -> prior -> decoder."""
    name, hps, model, oracle = setup
    b = _batch(hps, name)
    rf = hps.Common.final_reduction_factor
    pos_step = np.float32(model.mel_text_len_ratio) / np.float32(rf)
    text_embd = model.text_encoder(b["ids"], b["text_lengths"], pos_step=pos_step, training=False)
    pred = model.length_predictor(text_embd, b["text_lengths"], training=False).numpy()
    pred_ml = pred.astype(np.int32)
    reduced = (pred_ml + 80 + rf - 1) // rf
    z, _ = model.prior.sample(reduced, text_embd, b["text_lengths"], training=False, temperature=0.0)
    _, outs, _ = model.decoder(z, text_embd, reduced, b["text_lengths"], training=False, reduction_factor=rf)
    rmel, rlen, _ = oracle.test_step(b["ids"], b["text_lengths"])
    rf64 = oracle.last["pred_float"]
    margin = np.minimum(rf64 - np.floor(rf64), np.ceil(rf64) - rf64)
    print(f"[{name}] predicted frames {pred_ml + 80} oracle {rlen} integer margins {margin}")
    assert np.all(margin > 1e-4), "fixture sits on an integer boundary; pick another seed"
    assert np.array_equal(pred_ml + 80, rlen)            # bit-exact integer frame counts
    assert outs.shape == rmel.shape
    np.testing.assert_allclose(outs.numpy(), rmel, atol=MEL_TOL, rtol=0)


def test_posterior(setup):
    name, hps, model, oracle = setup
    b = _batch(hps, name)
    r = np.random.Generator(np.random.PCG64(3))
    reduced = (b["mel_lengths"] + 1) // 2
    Tz = int(reduced.max())
    mels = r.standard_normal((len(reduced), Tz, hps.Audio.num_mels)).astype(np.float32)
    mem = r.standard_normal((len(reduced), b["ids"].shape[1], hps.Encoder.Transformer.pre_hidden)).astype(np.float32)
    mu, logvar, _ = model.posterior(mels, mem, src_lengths=b["text_lengths"], target_lengths=reduced, training=False)
    rmu, rlv = oracle.posterior(mels.astype(np.float64), mem.astype(np.float64), b["text_lengths"], reduced)
    np.testing.assert_allclose(mu.numpy(), rmu, atol=1e-4, rtol=0)
    np.testing.assert_allclose(logvar.numpy(), rlv, atol=1e-4, rtol=0)


def test_prior_log_probability(setup):
    """prior.log_probability (prior.py:119-152): flow backwards; also log p(sample(eps)) == sample's own log-prob."""
    name, hps, model, oracle = setup
    b = _batch(hps, name)
    r = np.random.Generator(np.random.PCG64(11))
    reduced = (b["mel_lengths"] + 1) // 2
    Tz = int(reduced.max())
    z = r.standard_normal((len(reduced), Tz, hps.Common.latent_dim)).astype(np.float32)
    mem = r.standard_normal((len(reduced), b["ids"].shape[1], hps.Encoder.Transformer.pre_hidden)).astype(np.float32)
    got = model.prior.log_probability(z, mem, reduced, b["text_lengths"]).numpy()
    ref = oracle.prior_log_probability(z.astype(np.float64), mem.astype(np.float64), reduced, b["text_lengths"])
    np.testing.assert_allclose(got, ref, rtol=3e-5, atol=2e-2)
    zs, lp = model.prior.sample(reduced, mem, b["text_lengths"], eps=b["eps"])
    lp2 = model.prior.log_probability(zs, mem, reduced, b["text_lengths"]).numpy()
    np.testing.assert_allclose(lp2, lp.numpy(), rtol=1e-4, atol=5e-2)


@pytest.mark.parametrize("rf", [2, 5])
def test_elbo_forward(setup, rf):
    """VAENAR.call forward (models.py:105-197, training=False = dev_step train.py:148-155)."""
    name, hps, model, oracle = setup
    b = _batch(hps, name)
    r = np.random.Generator(np.random.PCG64(21))
    B, Tm = len(b["mel_lengths"]), int(b["mel_lengths"].max())
    Tz = (Tm + rf - 1) // rf
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((B, 1, Tz, hps.Common.latent_dim)).astype(np.float32)
    outs, l2, kl, ll, ali = model(b["ids"], mels, b["mel_lengths"], b["text_lengths"], reduction_factor=rf,
                                  training=False, reduce_loss=False, eps=eps)
    routs, rl2, rkl, rll, rali = oracle.call(b["ids"], mels, b["mel_lengths"], b["text_lengths"], rf, False, False,
                                             eps.astype(np.float64))
    np.testing.assert_allclose(outs.numpy(), routs, atol=MEL_TOL, rtol=0)
    np.testing.assert_allclose(l2.numpy(), rl2, rtol=1e-4)
    np.testing.assert_allclose(ll.numpy(), rll, rtol=1e-3, atol=1e-7)
    aux = model.last_aux.numpy()
    np.testing.assert_allclose(aux[1], oracle.last["post_lp"][:, 0], rtol=2e-5, atol=1e-2)
    np.testing.assert_allclose(aux[2], oracle.last["prior_lp"], rtol=3e-5, atol=3e-2)
    np.testing.assert_allclose(kl.numpy(), rkl, rtol=1e-3, atol=6e-2)      # difference of two O(1e3..1e4) log-probs
    for k in rali:
        np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)
    # reduce_loss=True: batch means (models.py:84,92,101)
    _, l2m, klm, llm, _ = model(b["ids"], mels, b["mel_lengths"], b["text_lengths"], reduction_factor=rf,
                                training=False, reduce_loss=True, eps=eps)
    np.testing.assert_allclose([l2m, llm], [rl2.mean(), rll.mean()], rtol=1e-3)


# ---- training-mode forward and data-dependent init (A17 training=True forward, A23) -------------
BN_PATHS = ["text_encoder/prenet/conv_stack/%d/bn/%s", "decoder/postnet/conv_stack/%d/bn/%s"]


def _fresh(name):
    """Training-mode calls mutate the weight store (BN moving statistics, ActNorm variables): private copies."""
    hps = tiny_hps() if name == "tiny" else LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic")
    model = VAENAR(hps, weights=w)
    oracle = Oracle(hps, {k: v.copy() for k, v in w.items()}, np.float64)
    oracle.update_moving_stats = True
    return hps, model, oracle


@pytest.mark.parametrize("name", ["tiny", "lj"])
def test_elbo_forward_training_mode(name):
    """VAENAR.call(training=True) forward (train.py:130-134): Dropout ON (counter-based masks reproduced bit for bit by
    the oracle), BatchNormalization on batch statistics, moving statistics updated."""
    hps, model, oracle = _fresh(name)
    try:
        b = _batch(hps, name)
        rf, seed = 2, 77
        r = np.random.Generator(np.random.PCG64(22))
        B, Tm = len(b["mel_lengths"]), int(b["mel_lengths"].max())
        Tz = (Tm + rf - 1) // rf
        mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
        eps = r.standard_normal((B, 1, Tz, hps.Common.latent_dim)).astype(np.float32)
        outs, l2, kl, ll, _ = model(b["ids"], mels, b["mel_lengths"], b["text_lengths"], reduction_factor=rf,
                                    training=True, reduce_loss=False, eps=eps, dropout_seed=seed)
        oracle.dropout_seed = seed
        routs, rl2, rkl, rll, _ = oracle.call(b["ids"], mels, b["mel_lengths"], b["text_lengths"], rf, True, False,
                                              eps.astype(np.float64))
        np.testing.assert_allclose(outs.numpy(), routs, atol=MEL_TOL, rtol=0)
        np.testing.assert_allclose(l2.numpy(), rl2, rtol=1e-4)
        np.testing.assert_allclose(ll.numpy(), rll, rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(kl.numpy(), rkl, rtol=1e-3, atol=6e-2)
        # dropout was really active: the eval-mode forward of the same inputs differs
        eouts, *_ = model(b["ids"], mels, b["mel_lengths"], b["text_lengths"], reduction_factor=rf, training=False,
                          reduce_loss=False, eps=eps)
        # (the eval forward above already runs on the UPDATED moving statistics: compare them with the oracle's)
        n_enc, n_post = hps.Encoder.Transformer.n_conv, hps.Decoder.Transformer.post_n_conv
        paths = [BN_PATHS[0] % (i, s) for i in range(n_enc) for s in ("moving_mean", "moving_variance")] + \
                [BN_PATHS[1] % (i, s) for i in range(n_post) for s in ("moving_mean", "moving_variance")]
        got = model.get_weights(paths)
        for p in paths:
            np.testing.assert_allclose(got[p], oracle.w[p], rtol=2e-5, atol=2e-6, err_msg=p)
        assert np.abs(eouts.numpy() - outs.numpy()).max() > 1e-3
        oracle.dropout_seed = None
        reouts, *_ = oracle.call(b["ids"], mels, b["mel_lengths"], b["text_lengths"], rf, False, False, eps.astype(np.float64))
        np.testing.assert_allclose(eouts.numpy(), reouts, atol=MEL_TOL, rtol=0)
    finally:
        model.engine.close()


@pytest.mark.parametrize("name", ["tiny", "lj"])
def test_model_init_actnorm(name):
    """VAENAR.init (models.py:212-226): data-dependent ActNorm init (flow.py:189-196) at max_reduction_factor."""
    hps, model, oracle = _fresh(name)
    try:
        b = _batch(hps, name)
        rf, seed = hps.Common.max_reduction_factor, 5
        red = (b["mel_lengths"].astype(np.int64) + rf - 1) // rf
        Tz, C = int(red.max()), hps.Common.latent_dim
        r = np.random.Generator(np.random.PCG64(23))
        eps = r.standard_normal((len(red), Tz, C)).astype(np.float32)
        mel = model.init(b["ids"], b["mel_lengths"], b["text_lengths"], eps=eps, dropout_seed=seed)
        oracle.dropout_seed = seed
        rmel = oracle.init(b["ids"], b["mel_lengths"], b["text_lengths"], eps.astype(np.float64))
        np.testing.assert_allclose(mel.numpy(), rmel, atol=MEL_TOL, rtol=0)
        paths = ["prior/glow/%d/0/%s" % (s, v) for s in range(hps.Prior.Transformer.n_blk) for v in ("log_scale", "bias")]
        got = model.get_weights(paths)
        for p in paths:
            np.testing.assert_allclose(got[p], oracle.w[p], rtol=1e-4, atol=2e-5, err_msg=p)
        # the re-packed engine now samples with the initialised flow: inference parity against the updated oracle
        oracle.dropout_seed = None
        bi = _batch(hps, name)
        gmel, _ = model.inference(bi["ids"], bi["mel_lengths"], bi["text_lengths"], eps=bi["eps"])
        omel, _ = oracle.inference(bi["ids"], bi["mel_lengths"], bi["text_lengths"], 2, bi["eps"].astype(np.float64))
        np.testing.assert_allclose(gmel.numpy(), omel, atol=MEL_TOL, rtol=0)
    finally:
        model.engine.close()


@pytest.mark.parametrize("rows64", [0, 1], ids=["rows32", "rows64"])
def test_s1_against_fp32_oracle(rows64):
    """The bench workload itself (B=16, T_text=128, T_mel=800, full lengths) against the fp32 NumPy oracle (seconds at this
    size), with the chain kernel's 32- and 64-row panels (the latter is what bench.py runs with several batches in flight)."""
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
    ref, rali = Oracle(hps, w, np.float32).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    model = VAENAR(hps, weights=w)
    try:
        model.engine.set_option("chain_rows64", rows64)
        model.engine.set_option("gemm_wide_tiles", rows64)
        mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        err = np.abs(mel.numpy() - ref).max()
        print(f"S1 rows64={rows64}: max-abs mel err vs fp32 oracle {err:.3e}")
        assert err < MEL_TOL
        for k in rali:
            np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)
    finally:
        model.engine.close()


def test_s1_both_chain_kernels_against_float64_oracle():
    """The two generations of the row-panel chain kernel on the bench workload against the float64 oracle: both at fp32 round-off
    level (3e-6).  The 4-wave kernel once sat at 1.3e-4: under fp contraction the compiler evaluated the high half of its fp16
    hi / lo split twice -- fused into the multiply for the value it subtracted, unfused for the value it stored -- and one element
    in ten thousand came out one fp16 ulp off (csrc/gemm3c.hip: split_hi_lo; profiles/r05_experiments.txt r05i) -- well inside the 1e-3 contract, invisible to every other test, hence this one."""
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
    ref, _ = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"].astype(np.float64))
    model = VAENAR(hps, weights=w)
    try:
        err = {}
        for w4 in (1, 0):
            model.engine.set_option("chain_waves4", w4)
            mel, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
            err[w4] = float(np.abs(mel.numpy() - ref).max())
        print("S1 max-abs mel err vs float64 oracle: 4-wave kernel %.3e, 8-wave kernel %.3e" % (err[1], err[0]))
        assert err[0] < 1e-5 and err[1] < 1e-5, err
    finally:
        model.engine.close()


def test_handles_in_flight_are_independent():
    """bench.py keeps several batches in flight on one GPU, one engine handle (= one HIP stream, workspace, weight copy) each,
    issued round-robin from one host thread without synchronising: every handle must return exactly what it returns alone."""
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    batches = [make_batch(4, 60 + 9 * i, 200 + 40 * i, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                          ragged=True, seed=50 + i, temperature=1.0, text_step=5, mel_step=13) for i in range(3)]
    models = [VAENAR(hps, weights=w) for _ in range(3)]
    try:
        for m in models:
            m.engine.set_option("chain_rows64", 1)
        alone = []
        for m, b in zip(models, batches):
            mel, _ = m.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
            alone.append(mel.numpy())                      # .numpy() synchronises: strictly one after another
        outs = []
        for rep in range(4):                               # 12 asynchronous calls dealt to the three handles
            for m, b in zip(models, batches):
                outs.append(m.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])[0])
        for i, o in enumerate(outs):
            np.testing.assert_array_equal(o.numpy(), alone[i % 3])
    finally:
        for m in models:
            m.engine.close()


def test_full_size_s1_properties():
    """BASELINE.json's full configuration (B=16, T_text=128, T_mel=800, rf=2) through size-independent properties (the
    float64 oracle takes minutes at this size): utterances are independent, so (i) a sub-batch run alone reproduces its rows
    of the full batch, (ii) permuting the batch permutes the output, (iii) ragged lengths only change the padded tail of
    each utterance's neighbours -- never another utterance; and the alignments are row-stochastic."""
    hps = LJHPS
    model = VAENAR(hps, weights=init_weights(hps, seed=1234, mode="synthetic", include_posterior=False))
    try:
        b = make_batch(16, 128, 800, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                       ragged=True, seed=1234, temperature=1.0, text_step=4, mel_step=24)
        mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        mel = mel.numpy()
        assert mel.shape == (16, 800, 80) and np.isfinite(mel).all()
        # (i) rows 3..6 alone (their own padded maximum is shorter: compare the frames that exist in both runs)
        sel = slice(3, 7)
        Tm = int(b["mel_lengths"][sel].max()); Tz = (Tm + 1) // 2
        sub, _ = model.inference(b["ids"][sel], b["mel_lengths"][sel], b["text_lengths"][sel], reduction_factor=2,
                                 eps=b["eps"][sel, :Tz])
        sub = sub.numpy()
        for i, row in enumerate(range(3, 7)):
            n = int(b["mel_lengths"][row]) - 12           # PostNet (5 convs of width 5) reaches 10 frames back from the padding
            np.testing.assert_allclose(sub[i, :n], mel[row, :n], atol=2e-5, rtol=0)
        # (ii) a batch permutation permutes the rows (same shapes -> same kernels -> tight tolerance)
        perm = np.random.Generator(np.random.PCG64(2)).permutation(16)
        pmel, _ = model.inference(b["ids"][perm], b["mel_lengths"][perm], b["text_lengths"][perm], reduction_factor=2,
                                  eps=b["eps"][perm])
        np.testing.assert_allclose(pmel.numpy(), mel[perm], atol=1e-6, rtol=0)
        # alignments: rows of valid queries sum to one over the valid keys
        a = ali["decoder-attention-1"].numpy()
        assert a.shape == (16, 4, 400, 128)
        np.testing.assert_allclose(a.sum(-1), 1.0, atol=1e-5)
        tl = int(b["text_lengths"][5])
        assert np.abs(a[5, :, :int((b["mel_lengths"][5] + 1) // 2), tl:]).max() == 0.0      # masked keys get exactly zero weight
    finally:
        model.engine.close()


@pytest.mark.parametrize("name", ["tiny", "lj"])
def test_self_attention_block_alone(name):
    """SelfAttentionBLK (attention.py:392-415) as a block: the same encoder with n_blk = 0 and n_blk = 1 on the same variables.  The
    n_blk = 0 engine's output IS the block's input (prenet -> projection + PE); the oracle's self_attention_blk applied to exactly
    that input must give the n_blk = 1 engine's output -- attention core over the key AND query masks, att_proj on concat(x, att),
    LayerNorm, FFN, LayerNorm, with nothing of the rest of the encoder in the comparison.  (lj: width 512, the LayerNorm-after-GEMM
    path; tiny: width 96, LayerNorm in the GEMM epilogue.)"""
    import copy
    hps1 = copy.deepcopy(tiny_hps() if name == "tiny" else LJHPS)
    hps1.Encoder.Transformer.n_blk = 1
    hps0 = copy.deepcopy(hps1)
    hps0.Encoder.Transformer.n_blk = 0
    w1 = init_weights(hps1, seed=77, mode="synthetic", include_posterior=False)
    w0 = {k: v for k, v in w1.items() if "/self_attentions/" not in k}
    b = make_batch(3, 37 if name == "lj" else 13, 40, vocab_size=hps1.Encoder.Transformer.vocab_size, latent_dim=hps1.Common.latent_dim,
                   ragged=True, text_step=6 if name == "lj" else 4, mel_step=7)
    m0, m1 = VAENAR(hps0, weights=w0), VAENAR(hps1, weights=w1)
    try:
        x = m0.text_encoder(b["ids"], b["text_lengths"], pos_step=2.0).numpy()
        y = m1.text_encoder(b["ids"], b["text_lengths"], pos_step=2.0).numpy()
    finally:
        m0.engine.close(); m1.engine.close()
    e = hps1.Encoder.Transformer
    ref = Oracle(hps1, w1, np.float64).self_attention_blk("text_encoder/self_attentions/0", x.astype(np.float64), b["text_lengths"],
                                                          e.attention_heads, e.attention_temperature)
    ref = ref[0] if isinstance(ref, tuple) else ref
    assert np.abs(y - x).max() > 0.1                                   # the block does something
    assert np.abs(y - ref).max() < 5e-5, np.abs(y - ref).max()


def test_s1_elbo_forward_against_float64_oracle():
    """VERDICT round 5 "next round" #3 (ii): the ELBO forward (VAENAR.call, models.py:105-197) at the bench SIZE (B = 16, T_text = 128,
    T_mel = 800, rf = 2, ragged) against the float64 oracle at fp32 round-off tolerances -- decoded mels 1e-5 absolute, the three
    per-utterance losses 1e-5 relative (the KL as the difference of two log-probabilities of order 1e5: relative to those).  The
    inference twin of this test (above) is what found the round-5 split defect; the posterior, the reparameterisation, both
    log-probabilities and the L2 terms had no test of this class."""
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic")
    b = make_batch(16, 128, 800, ragged=True, seed=4321, text_step=3, mel_step=17)
    r = np.random.Generator(np.random.PCG64(31))
    B, Tm = 16, 800
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((B, 1, Tm // 2, hps.Common.latent_dim)).astype(np.float32)
    oracle = Oracle(hps, w, np.float64)
    routs, rl2, rkl, rll, rali = oracle.call(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, False, False, eps.astype(np.float64))
    model = VAENAR(hps, weights=w)
    try:
        outs, l2, kl, ll, ali = model(b["ids"], mels, b["mel_lengths"], b["text_lengths"], reduction_factor=2, training=False,
                                      reduce_loss=False, eps=eps)
        aux = model.last_aux.numpy()
        e_mel = float(np.abs(outs.numpy() - routs).max())
        e_l2 = float(np.abs(l2.numpy() / rl2 - 1).max())
        e_ll = float(np.abs(ll.numpy() - rll).max() / max(1e-30, np.abs(rll).max()))
        lp_scale = np.maximum(np.abs(oracle.last["post_lp"][:, 0]), np.abs(oracle.last["prior_lp"]))
        e_post = float((np.abs(aux[1] - oracle.last["post_lp"][:, 0]) / lp_scale).max())
        e_prior = float((np.abs(aux[2] - oracle.last["prior_lp"]) / lp_scale).max())
        e_kl = float((np.abs(kl.numpy() - rkl) / lp_scale).max())
        e_ali = max(float(np.abs(ali[k].numpy() - rali[k]).max()) for k in rali)
        print(f"S1 ELBO forward vs float64: mel {e_mel:.2e}, l2 rel {e_l2:.2e}, length rel {e_ll:.2e}, posterior lp rel {e_post:.2e}, "
              f"prior lp rel {e_prior:.2e}, kl (relative to the log-probs, {lp_scale.max():.3g}) {e_kl:.2e}, alignments {e_ali:.2e}")
        assert e_mel < 1e-5 and e_l2 < 1e-5 and e_ll < 1e-5 and e_post < 1e-5 and e_prior < 1e-5 and e_kl < 1e-5 and e_ali < 1e-5
    finally:
        model.engine.close()
