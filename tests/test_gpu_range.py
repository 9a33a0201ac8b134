"""The arithmetic contract of the split-fp16 path (include/vaenar_hip.h, "Arithmetic contract of the split path").

TensorFlow evaluates every Dense / Conv1D / matmul of the path in fp32 (/root/reference/modules/attention.py:217-246,
modules/utils.py:44-53,76-85): activations of 1e5 or 1e-5 are ordinary there.  The engine's default path splits activations into
UNSCALED fp16 hi/lo pairs, which keeps fp32-class accuracy only inside a magnitude window.  These tests pin
  (i)   the exact-fp32 mode (`split_fp16 = 0`) at model level against the float64 oracle and the reference-on-shim fixtures;
  (ii)  the range guard: a model whose weights are rescaled so that a GEMM input tensor reaches 1e5 or 1e-5 -- mathematically the
        same function, which the oracle and the reference evaluate without trouble -- must still match (the engine surveys the
        ranges on the first call and keeps those modules on exact fp32 MFMA), an in-window model must stay on the split path,
        out-of-window attention operands run on the exact mode's scaled cores, and (round 6) the range SENTINEL catches what the
        survey cannot see -- inputs of a LATER call, a training step -- without ever handing out a non-finite number.
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
MEL_TOL = 2e-5          # asserted (round 6: was 2e-4); the contract is 1e-3 (north star)


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
        model.engine.set_option("range_guard", 0)          # (forgets the survey's states: the modules are back on the split path)
        model.engine.set_option("range_sentinel", 0)       # ... unwatched
        raw, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        bad = raw.numpy()
        unguarded = np.inf if not np.isfinite(bad).all() else np.abs(bad - ref).max()
        print(f"unguarded split path on the rescaled weights: max-abs mel err {unguarded:.3e}")
        assert unguarded > 20 * np.abs(got - ref).max()
        if factor > 1:
            # the survey off, the sentinel on: the overflow is caught at the checkpoint and the call replayed on exact fp32 (round 6)
            model.engine.set_option("range_sentinel", 1)
            mel4, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
            got4 = mel4.numpy()
            i4 = model.engine.range_info()
            print("survey off, sentinel on:", i4)
            assert np.isfinite(got4).all() and np.abs(got4 - ref).max() < MEL_TOL and i4["sentinel_trips"] == 1 and i4["replays"] == 1
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


def test_out_of_window_attention_operands_run_on_scaled_cores():
    """Query kernel x 2^18, key kernel / 2^18 in a decoder block: the logits are unchanged, but the query is of order 1e5 and the key of
    order 1e-6.  Round 5 refused such weights (its attention cores split Q, K, V unscaled in every mode); since round 6 the survey moves
    the modules to the exact mode, whose cores take per-launch power-of-two operand scales (attention2.hip: AttnArgs::qkv_absmax) -- the
    call must succeed and match the float64 oracle of the RESCALED weights, alignments included (attention.py:224-246 has no window)."""
    hps = LJHPS
    w0 = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    p = "decoder/attentions/0/cross_attention/"
    w = _rescaled(w0, [(p + "query_layer/kernel", 2.0 ** 18), (p + "key_layer/kernel", 2.0 ** -18)])
    b = _batch(hps, "lj")
    ref, rali = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    model = VAENAR(hps, weights=w)
    try:
        mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        got = mel.numpy()
        info = model.engine.range_info()
        print("scaled attention cores:", info)
        assert np.isfinite(got).all() and info["decoder"] == 2 and info["hi"] >= 32768.0
        assert np.abs(got - ref).max() < MEL_TOL
        for k in rali:
            np.testing.assert_allclose(ali[k].numpy(), rali[k], atol=1e-5, rtol=0)
    finally:
        model.engine.close()


@pytest.mark.parametrize("scale,trips", [(8.0e3, None), (3.0e4, True)], ids=["z-past-2^15", "z-past-65504"])
def test_sentinel_catches_inputs_that_leave_the_range_on_a_later_call(scale, trips):
    """the survey is a sample of ONE call.  In-window weights, a first call with unit noise (survey: in window, split
    path), then a SECOND call whose injected noise is `scale` times larger: the latent z -- the input of every flow step's folded
    ActNorm o InvertibleLinear product and of the pre-projections -- starts at max |eps| ~ 3.6e4, past the survey window's 2^15 (inside
    fp16 as long as no flow step amplifies it: the split keeps its 22 bits; whether the sentinel trips there is printed, not asserted)
    resp. ~1.3e5, past 65504 (hi = inf: it must trip).  (Not larger: from |z| ~ 1e6 on the attention LOGITS of the un-normalised
    pre-projection fall below the reference's mask fill value -2^32 + 1, attention.py:240 -- there TensorFlow lets MASKED keys win the
    softmax, while every kernel here skips masked keys because their weight is exactly 0 above that point: a documented boundary of the
    port, not of the split.)  Nothing may come back non-finite: the sentinel trips at `.numpy()`, the binding
    replays the call on exact fp32, and the result matches both the engine's own exact mode (bit for bit: same kernels) and the float64
    oracle (TensorFlow's fp32 carries 1e6 like any other number, /root/reference/modules/flow.py:149-166, transform.py:45-51)."""
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    b = _batch(hps, "lj")
    big = (np.asarray(b["eps"], np.float64) * scale).astype(np.float32)
    print("max |eps| of the second call: %.3g" % np.abs(big).max())
    ref, _ = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, big)
    # fp32 itself loses ground at this magnitude (an O(1) shift added to a latent of 1e6 keeps 0.06 of absolute resolution): the yardstick
    # is what the SAME statement costs in plain fp32 arithmetic (the float32 NumPy restatement), not 1e-3 out of thin air
    ref32, _ = Oracle(hps, w, np.float32).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, big)
    exact = VAENAR(hps, weights=w)
    model = VAENAR(hps, weights=w)
    try:
        exact.engine.set_option("split_fp16", 0)
        want = exact.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=big)[0].numpy()
        mel1, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        first = mel1.numpy()
        i0 = model.engine.range_info()
        assert (i0["encoder"], i0["prior"], i0["decoder"], i0["sentinel_trips"]) == (1, 1, 1, 0)
        mel2, ali2 = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=big)     # asynchronous: nothing known yet
        got = mel2.numpy()                                   # the checkpoint
        i1 = model.engine.range_info()
        print("after the second call:", i1)
        assert np.isfinite(got).all()
        for k in ali2:
            assert np.isfinite(ali2[k].numpy()).all()
        scale_ref = max(1.0, float(np.abs(ref).max()))
        err = np.abs(got - ref).max() / scale_ref
        print(f"second call: max-abs mel err {err:.3e} (relative to max |mel| = {scale_ref:.3g}); exact-mode run differs by {np.abs(got - want).max():.3e}")
        err32 = np.abs(np.asarray(ref32, np.float64) - ref).max() / scale_ref
        print(f"  the float32 NumPy restatement of the same call: {err32:.3e}")
        assert err < max(1e-3, 3.0 * err32)                  # the north star's tolerance, or fp32's own round-off at this magnitude
        if trips:
            assert i1["sentinel_trips"] == 1 and i1["replays"] == 1
            assert i1["prior"] == 2 and i1["decoder"] == 2   # the modules of the flagged call stay on exact fp32
            assert np.array_equal(got, want)                 # the replay ran the exact mode's kernels
            # ... and the NEXT ordinary call is served from there too, still correct
            again = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])[0].numpy()
            assert np.abs(again - first).max() < MEL_TOL and model.engine.range_info()["sentinel_trips"] == 1
        elif i1["sentinel_trips"]:
            assert np.array_equal(got, want)
        # what the sentinel protects against: the same second call with it switched off
        raw_model = VAENAR(hps, weights=w)
        try:
            raw_model.engine.set_option("range_sentinel", 0)
            raw_model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
            bad = raw_model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=big)[0].numpy()
            unguarded = np.inf if not np.isfinite(bad).all() else np.abs(bad - ref).max() / scale_ref
            print(f"unwatched split path on the same call: {unguarded:.3e}")
            if trips:
                assert not unguarded < 1e-3
        finally:
            raw_model.engine.close()
    finally:
        model.engine.close(); exact.engine.close()


def test_sentinel_replays_a_chain_of_pending_module_calls_in_order():
    """The module-by-module path (models.py:201-209: text encoder -> prior.sample -> decoder): calls whose outputs feed each other are
    pending together when `.numpy()` reaches the checkpoint.  The flagged one is the prior (its latent leaves fp16's range); the decoder's
    first run consumed an invalid z, so the binding must replay BOTH, in order, and hand out what the exact mode computes.  With host
    arguments in between (`fused=False` uploads the lengths per module) every upload is a checkpoint of its own and the same end result
    must come out.  Also: more pending calls than the log holds force a checkpoint by themselves."""
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    b = _batch(hps, "lj")
    big = (np.asarray(b["eps"], np.float64) * 3.0e4).astype(np.float32)
    ref, _ = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, big)
    scale_ref = max(1.0, float(np.abs(ref).max()))
    pos_step = np.float32(hps.Common.mel_text_len_ratio) / np.float32(2)
    red = ((b["mel_lengths"].astype(np.int64) + 1) // 2).astype(np.int32)
    model = VAENAR(hps, weights=w)
    try:
        e = model.engine
        model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"], fused=False)[0].numpy()     # surveys: in window
        assert e.range_info()["sentinel_trips"] == 0 and not e._log
        # (a) device-resident arguments: nothing synchronises between the calls
        red_d, tl_d, eps_d = e.to_device(red, np.int32), e.to_device(b["text_lengths"], np.int32), e.to_device(big, np.float32)
        te = model.text_encoder(b["ids"], b["text_lengths"], pos_step=pos_step)
        te.numpy()                                            # clean checkpoint
        z, _ = model.prior.sample(red, te, tl_d, eps=eps_d)   # (host lengths: one upload = one clean checkpoint BEFORE the call)
        _, mel, _ali = model.decoder(inputs=z, text_embd=te, z_lengths=red_d, text_lengths=tl_d, reduction_factor=2)
        pending = [name for name, _ in e._log if name.startswith("vnr_prior") or name.startswith("vnr_decoder")]
        assert pending == ["vnr_prior_sample", "vnr_decoder_fwd"], [name for name, _ in e._log]
        got = mel.numpy()
        info = e.range_info()
        print("two pending calls replayed:", info)
        assert np.isfinite(got).all() and info["sentinel_trips"] == 1 and info["replays"] == 1 and not e._log
        assert info["prior"] == 2 and info["decoder"] == 2
        assert np.abs(got - ref).max() / scale_ref < 1e-3
        assert np.isfinite(z.numpy()).all()                   # the replay rewrote the intermediate too
    finally:
        model.engine.close()
    model = VAENAR(hps, weights=w)
    try:
        e = model.engine
        model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"], fused=False)[0].numpy()
        # (b) host arguments between the modules: the trip surfaces at an upload inside the decoder's call, is replayed there, same result
        mel, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=big, fused=False)
        got = mel.numpy()
        info = e.range_info()
        # (two trips here: the prior's surfaces at the upload inside the decoder's call and moves the PRIOR to the exact mode; the decoder then
        #  runs on the split path with the replayed, still huge z, and trips at `.numpy()` in its own right)
        print("host arguments between the modules:", info)
        assert np.isfinite(got).all() and info["sentinel_trips"] == 2 and info["replays"] == 2
        assert info["prior"] == 2 and info["decoder"] == 2
        assert np.abs(got - ref).max() / scale_ref < 1e-3
        # a log that fills up synchronises (and checks) by itself: no unbounded queue of unverified results
        e._log_cap = 4
        ids_d, tl_d = e.to_device(b["ids"], np.int32), e.to_device(b["text_lengths"], np.int32)
        te = None
        for _ in range(6):
            te = model.text_encoder(ids_d, tl_d, pos_step=pos_step)
        assert len(e._log) < 4
        assert np.isfinite(te.numpy()).all()
    finally:
        model.engine.close()


def test_sentinel_replay_repeats_the_options_set_in_front_of_the_call():
    """VAENAR.call with n_sample = 2 (models.py:141-178) sets the engine's sample count, issues vnr_elbo_fwd and restores it: a replay that
    started at the call would run it with the restored value on buffers sized for two samples.  Evaluation-mode ELBO forward whose mels
    are 1e6 times larger than the surveyed call's (the posterior's PreNet output leaves fp16's range): one trip, one replay, the same
    numbers as an engine that ran on the exact mode from the start (same kernels: bit for bit)."""
    hps = tiny_hps()
    hps.Train.num_samples = 2
    w = init_weights(hps, seed=1234, mode="synthetic")
    b = _batch(hps, "tiny")
    r = np.random.Generator(np.random.PCG64(17))
    B, Tm = len(b["mel_lengths"]), int(b["mel_lengths"].max())
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((B, 2, (Tm + 1) // 2, hps.Common.latent_dim)).astype(np.float32)
    huge = (mels.astype(np.float64) * 1.0e6).astype(np.float32)
    exact, model = VAENAR(hps, weights=w), VAENAR(hps, weights=w)
    try:
        exact.engine.set_option("split_fp16", 0)
        want = exact(b["ids"], huge, b["mel_lengths"], b["text_lengths"], reduction_factor=2, training=False, reduce_loss=False, eps=eps)
        want = [want[0].numpy(), want[1].numpy(), want[2].numpy(), want[3].numpy()]
        model(b["ids"], mels, b["mel_lengths"], b["text_lengths"], reduction_factor=2, training=False, reduce_loss=False, eps=eps)[0].numpy()
        assert model.engine.range_info()["sentinel_trips"] == 0
        outs, l2, kl, ll, _ = model(b["ids"], huge, b["mel_lengths"], b["text_lengths"], reduction_factor=2, training=False,
                                    reduce_loss=False, eps=eps)
        assert ("vnr_set_option", (b"n_sample", 2)) in model.engine._log                 # recorded in front of the pending call
        got = [outs.numpy(), l2.numpy(), kl.numpy(), ll.numpy()]
        info = model.engine.range_info()
        print("n_sample = 2 replay:", info)
        assert info["sentinel_trips"] == 1 and info["replays"] == 1
        assert got[0].shape == (2 * B, Tm, hps.Audio.num_mels)
        for g_, w_ in zip(got, want):
            assert np.isfinite(g_).all() and np.array_equal(g_, w_)
    finally:
        model.engine.close(); exact.engine.close()


def test_sentinel_c_abi_semantics_without_the_python_replay():
    """What a C caller sees (include/vaenar_hip.h): the flagged call itself returns VNR_OK (asynchronous); the next synchronisation point
    returns VNR_ERR_RANGE once, the modules are in state 2, and re-issuing the call gives the exact result."""
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    b = _batch(hps, "lj")
    big = (np.asarray(b["eps"], np.float64) * 3.0e5).astype(np.float32)
    model = VAENAR(hps, weights=w)
    try:
        e = model.engine
        model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])[0].numpy()
        mel, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=big)
        e._log.clear()                                        # play the C caller: no replay log
        rc = e.lib.vnr_synchronize(e.handle)
        assert rc == _lib.VNR_ERR_RANGE and b"exact fp32" in e.lib.vnr_last_error(e.handle)
        assert e.lib.vnr_synchronize(e.handle) == 0           # reported once
        assert e.range_info()["prior"] == 2
        mel, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=big)
        assert e.lib.vnr_synchronize(e.handle) == 0 and np.isfinite(mel.numpy()).all()
    finally:
        model.engine.close()


def test_survey_takes_one_record_per_attention_operand_at_batch_32():
    """the survey took 3 records per attention call AND batch element, so its 2048-entry table overflowed silently
    at B = 17 and the later modules were marked in-window unseen.  One batched launch per operand now; a full table is an error."""
    hps = tiny_hps()
    w = init_weights(hps, seed=1234, mode="synthetic")
    b = make_batch(32, 11, 40, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True,
                   temperature=1.0, text_step=1, mel_step=1)
    model, oracle = VAENAR(hps, weights=w), Oracle(hps, w, np.float64)
    try:
        mel, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        got = mel.numpy()
        info = model.engine.range_info()
        print("B = 32 survey:", info)
        assert info["surveys"] == 1 and (info["encoder"], info["prior"], info["decoder"]) == (1, 1, 1)
        ref, _ = oracle.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
        assert np.abs(got - ref).max() < MEL_TOL
    finally:
        model.engine.close()


def test_sentinel_in_the_training_step():
    """A training step whose mels are 1e6 times larger than anything before (the posterior's PreNet output, a Dense input, leaves fp16's
    range): the forward's split products raise the sentinel, Adam and the BatchNormalization moving updates are predicated on it, and
    vnr_train_step repeats the step on exact fp32 MFMA with scaled attention cores and fp32 kernel-gradient GEMMs.  Nothing non-finite may
    reach the variables, and the outcome must equal a handle that ran on that path from the start."""
    hps = tiny_hps()
    w = init_weights(hps, seed=1234, mode="synthetic")
    r = np.random.Generator(np.random.PCG64(9))
    B, Tt, Tm = 3, 11, 40
    ids = r.integers(1, hps.Encoder.Transformer.vocab_size, (B, Tt)).astype(np.int32)
    tl = np.array([11, 9, 7], np.int32); ml = np.array([40, 33, 26], np.int32)
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((B, 1, Tm // 2, hps.Common.latent_dim)).astype(np.float32)
    huge = (mels.astype(np.float64) * 1.0e6).astype(np.float32)
    probe = ["posterior/prenet/dense1/kernel", "decoder/out_projection/kernel", "text_encoder/prenet/conv_stack/0/bn/moving_mean"]

    def run(force_fp32):
        m = VAENAR(hps, weights=w)
        try:
            m.engine.set_option("deterministic", 1)
            if force_fp32:
                m.engine.set_option("train_fp32", 1)
            out = [m.train_step(ids, mels, tl, ml, 1.0, 2, eps=eps, dropout_seed=3)]
            out.append(m.train_step(ids, huge, tl, ml, 1.0, 2, eps=eps, dropout_seed=4))
            out.append(m.train_step(ids, mels, tl, ml, 1.0, 2, eps=eps, dropout_seed=5))
            info = m.engine.range_info()
            have = [k for k in probe if k in m.engine._weights_loaded]
            return out, {k: v for k, v in m.get_weights(have).items()}, info, m.engine.get_optimizer_step()
        finally:
            m.engine.close()

    out, wts, info, steps = run(False)
    print("losses:", out, info)
    assert all(np.isfinite(o).all() for o in out) and steps == 3
    assert all(np.isfinite(v).all() for v in wts.values())
    assert info["train_fp32"] == 1 and info["sentinel_trips"] == 1
    # the same three steps with the middle and last one forced onto the exact path from the start of step 2 cannot be arranged from outside;
    # a handle on the exact path throughout differs from ours only in step 1 (split against exact products: fp32 round-off class)
    out32, wts32, info32, _ = run(True)
    assert info32["sentinel_trips"] == 0
    np.testing.assert_allclose(np.asarray(out[1]), np.asarray(out32[1]), rtol=2e-3)
    np.testing.assert_allclose(np.asarray(out[2]), np.asarray(out32[2]), rtol=2e-3)
    lr = hps.Train.learning_rate
    for k in wts:
        if "moving_" in k:
            # the text encoder's batch statistics do not depend on the mels: both handles must hold the SAME moving statistics -- the first,
            # flagged attempt of step 2 did not update them (its update was predicated on the sentinel), the repeat did, once
            np.testing.assert_allclose(wts[k], wts32[k], rtol=1e-4, atol=1e-6, err_msg=k)
        else:
            # Adam moves an entry by at most ~lr per step whatever the gradient's size (near-zero gradients flip sign on round-off, so the two
            # handles need not agree entry by entry): three sane steps, no more
            assert np.abs(wts[k] - np.asarray(w[k], np.float32)).max() <= 3.5 * lr, k
            assert np.abs(wts32[k] - np.asarray(w[k], np.float32)).max() <= 3.5 * lr, k
