"""Round-2 GPU tests: the BASELINE.json configurations no `-m gpu` test exercised before (T1 at full size, an LJ-sized B=1
run of the inference.py harness, the DataBaker model), the device noise generator, the checkpoint / resume loop with
optimizer state, and a population test of the integer frame counts."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from vaenar_tts_amd.configs import DataBakerHPS, LJHPS, tiny_hps
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights, is_trainable

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- device noise (prior.py:35, posterior.py:35) -----------------------------------------------------------------------------
def test_device_normal_matches_the_oracle_generator():
    from oracle.vaenar_numpy import philox_normal
    hps = tiny_hps()
    model = VAENAR(hps, weights=init_weights(hps, seed=1))
    try:
        eng = model.engine
        for n, seed, off, sd in ((4096, 1234, 0, 1.0), (1003, 2 ** 40 + 17, 2 ** 33 + 5, 0.667), (1, 7, 0, 1.0)):
            got = eng.random_normal((n,), seed, off, sd).numpy()
            ref = philox_normal(n, seed, off, sd)
            # same integers, same Box-Muller; fp32 log / sincos on the device vs float64 in the oracle
            np.testing.assert_allclose(got, ref, atol=4e-6 * sd, rtol=2e-6)
        z = eng.random_normal((16, 400, 128), 99).numpy().ravel()        # one S1 noise tensor
        assert abs(z.mean()) < 3e-3 and abs(z.std() - 1) < 3e-3 and abs((z ** 4).mean() - 3) < 3e-2
        # successive draws of the module take disjoint counter ranges; re-seeding replays
        model.prior.seed(5)
        a = model.prior.draw((3, 10, hps.Common.latent_dim)).numpy()
        b = model.prior.draw((3, 10, hps.Common.latent_dim)).numpy()
        model.prior.seed(5)
        assert np.array_equal(model.prior.draw((3, 10, hps.Common.latent_dim)).numpy(), a) and not np.array_equal(a, b)
        assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 0.2
    finally:
        model.engine.close()


def test_temperature_sampling_uses_the_device_stream():
    """inference with temperature > 0 and no injected eps: noise is drawn on the device; the same seed reproduces the mel,
    and the result equals an injected-eps run with the oracle's restatement of the generator."""
    from oracle.vaenar_numpy import philox_normal
    hps = tiny_hps()
    w = init_weights(hps, seed=3, mode="synthetic", include_posterior=False)
    b = make_batch(2, 9, 24, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, text_step=2, mel_step=4)
    model = VAENAR(hps, weights=w)
    try:
        model.prior.seed(77)
        m1, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], temperature=0.5)
        model.prior.seed(77)
        m2, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], temperature=0.5)
        assert np.array_equal(m1.numpy(), m2.numpy())
        Tz = (int(b["mel_lengths"].max()) + 1) // 2
        eps = philox_normal(2 * Tz * hps.Common.latent_dim, 77, 0, 0.5).reshape(2, Tz, hps.Common.latent_dim)
        m3, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], eps=eps)
        assert np.abs(m3.numpy() - m1.numpy()).max() < 1e-4
    finally:
        model.engine.close()


# ---- checkpoint / resume (train.py:246-255, inference.py:122-123) ---------------------------------------------------------------
def _train_case(hps, B=3, Tt=11, Tm=40, seed=31):
    b = make_batch(B, Tt, Tm, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, text_step=2, mel_step=5)
    r = np.random.Generator(np.random.PCG64(seed))
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((B, (Tm + 1) // 2, hps.Common.latent_dim)).astype(np.float32)
    return b, mels, eps


def test_resume_from_tensor_bundle_continues_adam(tmp_path):
    """train 2 steps -> CheckpointManager.save (TensorFlow tensor bundle with variables + Adam slots + counters) -> NEW engine
    -> restore -> step 3 equals the uninterrupted run's step 3; without the optimizer state it does not."""
    from vaenar_tts_amd.tf_checkpoint import CheckpointManager, list_variables
    hps = tiny_hps()
    w = init_weights(hps, seed=1234, mode="synthetic")
    b, mels, eps = _train_case(hps)
    args = (b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2)
    a = VAENAR(hps, weights=w)
    try:
        for s in (1, 2):
            a.train_step(*args, eps=eps, dropout_seed=s, learning_rate=1e-3)
        mgr = CheckpointManager(str(tmp_path), max_to_keep=20)
        prefix = mgr.save(lambda p, n: a.save_checkpoint(p, step=5, save_counter=n))
        a.train_step(*args, eps=eps, dropout_seed=3, learning_rate=1e-3)
        want = a.get_weights()
        m_a, v_a, it_a = a.get_optimizer_state()
    finally:
        a.engine.close()
    assert it_a == 3 and os.path.basename(prefix) == "ckpt-1" and os.path.exists(prefix + ".index")
    keys = {k for k, _, _ in list_variables(prefix)}
    assert "optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE" in keys and "step/.ATTRIBUTES/VARIABLE_VALUE" in keys
    assert "model/decoder/pre_projection/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE" in keys

    bm = VAENAR(hps, weights=init_weights(hps, seed=999, mode="synthetic"))       # different weights: everything must come from the file
    try:
        step = bm.restore_checkpoint(CheckpointManager(str(tmp_path)).latest_checkpoint)
        assert step == 5 and bm.engine.get_optimizer_step() == 2
        bm.train_step(*args, eps=eps, dropout_seed=3, learning_rate=1e-3)
        got = bm.get_weights()
        m_b, v_b, it_b = bm.get_optimizer_state()
    finally:
        bm.engine.close()
    assert it_b == 3
    worst = max(float(np.abs(got[k] - want[k]).max()) for k in want)
    # the kernel-gradient GEMMs accumulate with float atomics (summation order varies run to run): equal to rounding, not bitwise
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=2e-5, atol=2e-6, err_msg=k)
    for k in m_a:
        np.testing.assert_allclose(m_b[k], m_a[k], rtol=1e-4, atol=1e-9, err_msg=k)
        np.testing.assert_allclose(v_b[k], v_a[k], rtol=1e-4, atol=1e-12, err_msg=k)

    c = VAENAR(hps, weights=init_weights(hps, seed=999, mode="synthetic"))        # variables only (what round 1 did): Adam restarts
    try:
        c.load_weights(prefix)
        c.train_step(*args, eps=eps, dropout_seed=3, learning_rate=1e-3)
        cold = c.get_weights()
    finally:
        c.engine.close()
    k = "decoder/pre_projection/kernel"
    assert np.abs(cold[k] - want[k]).max() > 50 * max(worst, 1e-7)               # bias correction t=1 vs t=3: a visibly different step


def test_checkpoint_prefix_into_inference(tmp_path):
    """inference.py:122-123: tf.train.Checkpoint(model=model).restore(prefix).expect_partial() -> VAENAR.load_weights(prefix)
    on a bundle that also holds optimizer entries -> inference equals the engine that wrote it."""
    hps = tiny_hps()
    w = init_weights(hps, seed=4, mode="synthetic")
    b = make_batch(2, 9, 24, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, text_step=2, mel_step=4,
                   temperature=1.0)
    a = VAENAR(hps, weights=w)
    try:
        prefix = a.save_checkpoint(str(tmp_path / "ckpt-7"), step=6, save_counter=7)
        want, _ = a.inference(b["ids"], b["mel_lengths"], b["text_lengths"], eps=b["eps"])
        want = want.numpy()
    finally:
        a.engine.close()
    c = VAENAR(hps)
    try:
        c.load_weights(prefix)
        got, _ = c.inference(b["ids"], b["mel_lengths"], b["text_lengths"], eps=b["eps"])
        assert np.array_equal(got.numpy(), want)
    finally:
        c.engine.close()


# ---- BASELINE config 1: one LJ-sized utterance through the inference.py harness (batch_size = 1) ------------------------------
def test_inference_harness_lj_single_utterance(tmp_path):
    from oracle.vaenar_numpy import Oracle
    out = subprocess.run([sys.executable, os.path.join(ROOT, "inference.py"), "--dataset", "ljspeech", "--batch_size", "1",
                          "--num_utterances", "2", "--test_dir", str(tmp_path), "--seed", "1234"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Average RTF is" in out.stdout
    hps = LJHPS
    rf = hps.Common.final_reduction_factor
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    data = make_batch(2, 96, 2 * rf, vocab_size=hps.Encoder.Transformer.vocab_size, ragged=True, seed=1234, text_step=5)
    orc = Oracle(hps, w, np.float64)
    for fid in range(2):
        got = np.load(os.path.join(str(tmp_path), "prior-%d-0.npy" % fid))
        tl = data["text_lengths"][fid:fid + 1]
        ids = data["ids"][fid:fid + 1, :int(tl.max())]
        mel, pl80, _ = orc.test_step(ids, tl)                    # test_step arithmetic of inference.py:129-143 on the oracle
        pred = orc.last["pred_float"]
        assert np.abs(pred - np.round(pred)).min() > 1e-3        # the truncation is not decided by rounding noise
        assert got.shape == (int(pl80[0]), hps.Audio.num_mels) and got.dtype == np.float32
        assert np.abs(got - mel[0, :got.shape[0]]).max() < 2e-5


# ---- BASELINE config 3's harness: train.py end to end (init step, schedules, dev pass, checkpoints, resume), deterministic by default ---------
def test_train_harness_runs_resumes_and_is_reproducible(tmp_path):
    """The repo-root train.py (counterpart of /root/reference/train.py:114-306) on the tiny configuration: initial checkpoint + init step,
    two epochs of two steps with the KL-weight and reduction-factor schedules, dev pass, a checkpoint per epoch (TensorFlow tensor
    bundles + the `checkpoint` state file).  A second directory with the same seed prints the SAME losses to the last digit (the
    harness turns the engine's deterministic mode on, like the reference's TF_DETERMINISTIC_OPS=1); a third invocation on the first
    directory restores the last checkpoint and goes on."""
    def run(model_dir, epochs):
        cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--dataset", "tiny", "--model_dir", str(model_dir), "--log_dir", str(tmp_path / "log"),
               "--epochs", str(epochs), "--steps_per_epoch", "2", "--batch_size", "4", "--t_text", "12", "--t_mel", "40", "--seed", "77"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout

    def losses(text):           # the per-step lines without their wall-clock field
        return [ln.rsplit(", time", 1)[0] for ln in text.splitlines() if ln.startswith("Step ") or ln.startswith("Initial step")]

    a = run(tmp_path / "a", 2)
    assert "Initializing from scratch." in a and "Initial checkpoint for step 0" in a and "Training Epoch 2" in a and "dev-total" in a
    la = losses(a)
    assert len(la) == 1 + 2 * 2 and all(np.isfinite([float(x) for x in __import__("re").findall(r"-?\d+\.\d+", ln)]).all() for ln in la)
    files = sorted(os.listdir(tmp_path / "a"))
    assert "checkpoint" in files and any(f.startswith("ckpt-3.index") for f in files), files      # initial + two epochs
    b = run(tmp_path / "b", 2)
    assert losses(b) == la, (losses(b), la)                      # same seed, same bits
    c = run(tmp_path / "a", 3)
    # (the reference saves and THEN increments its epoch counter, train.py:300-303: the stored counter lags by one and a restart repeats
    #  the last finished epoch -- reproduced, not repaired)
    assert "Restored from" in c and "Initial step" not in c and "Training Epoch 1," not in c and "Training Epoch 2," in c and "Training Epoch 3," in c
    assert len(losses(c)) == 4


def test_training_soak_changing_shapes_memory_steady():
    """60 optimizer steps at LJ widths per mode (float atomics / deterministic) with batch shapes and reduction factors changing from step
    to step (tools/train_soak.py): every loss finite, the loss falls, and free device memory is the same at the last checkpoint as at the
    third -- the workspace arena, the activation-gradient chunks, the deterministic scratch and the event pool reach a steady state."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_soak.py"), "60"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "soak ok" in out.stdout and out.stdout.count("non-finite 0") == 2
    first = [float(ln.split("loss")[1].split()[0]) for ln in out.stdout.splitlines() if " step   0 " in ln]
    last = [float(ln.split("loss")[1].split()[0]) for ln in out.stdout.splitlines() if " step  59 " in ln]
    assert len(first) == 2 and len(last) == 2 and all(b < 0.6 * a for a, b in zip(first, last)), (first, last)


# ---- BASELINE config 5's model: DataBakerHPS (vocab 39, mel/text ratio 4.21) ------------------------------------------------
def test_databaker_inference_and_train_step():
    from oracle import kinks
    from oracle.vaenar_numpy import Oracle
    hps = DataBakerHPS
    assert hps.Encoder.Transformer.vocab_size == 39 and abs(hps.Common.mel_text_len_ratio - 4.21) < 1e-9
    w = init_weights(hps, seed=1234, mode="synthetic")
    b = make_batch(2, 21, 50, vocab_size=39, latent_dim=hps.Common.latent_dim, ragged=True, text_step=4, mel_step=8, temperature=1.0)
    r = np.random.Generator(np.random.PCG64(5))
    mels = r.standard_normal((2, 50, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((2, 25, hps.Common.latent_dim)).astype(np.float32)
    model = VAENAR(hps, weights=w)
    try:
        assert abs(model.mel_text_len_ratio - 4.21) < 1e-9
        mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], eps=b["eps"])
        ref, rali = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
        assert np.abs(mel.numpy() - ref).max() < 2e-5
        for k in rali:
            assert np.abs(ali[k].numpy() - rali[k]).max() < 1e-4
        loss, mel_l2, kl, len_l2 = model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2, eps=eps, dropout_seed=4,
                                                    apply_update=False)
        got = model.gradients()
    finally:
        model.engine.close()
    sc, flipped = kinks.compare(got, kinks.torch_oracle_run(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, 1.0, 4))
    assert abs(loss - sc["loss"]) < 1e-4 * max(1, abs(sc["loss"])) and len(flipped) <= 3


def test_databaker_data_parallel_rank_shape_with_rccl_bound():
    """BASELINE config 5 as ONE rank sees it: DataBakerHPS, global batch 32 over 8 ranks = 4 utterances per rank, the RCCL communicator
    bound (one rank here: no multi-GPU box), deterministic accumulation as train.py runs it.  Every gradient against the float64
    autograd restatement; the communicator reports its own size and rank; the all-reduced gradient of a one-rank job is the local one."""
    from oracle import kinks
    hps = DataBakerHPS
    B, Tt, Tm = 4, 24, 96                                        # (mel / text ratio 4.21 -> ~4 frames per token)
    w = init_weights(hps, seed=77, mode="synthetic")
    b = make_batch(B, Tt, Tm, vocab_size=39, latent_dim=hps.Common.latent_dim, ragged=True, seed=5, text_step=3, mel_step=7)
    r = np.random.Generator(np.random.PCG64(9))
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((B, Tm // 2, hps.Common.latent_dim)).astype(np.float32)
    model = VAENAR(hps, weights=w)
    try:
        model.engine.set_option("deterministic", 1)
        args = (b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2)
        model.train_step(*args, eps=eps, dropout_seed=11, apply_update=False)
        g_local = model.gradients()
        model.engine.comm_init(1, 0, model.engine.comm_unique_id())
        assert model.engine.comm_info() == (1, 0)                # ncclCommCount / ncclCommUserRank of the bound communicator
        model.engine.comm_broadcast_weights()
        loss, mel_l2, kl, len_l2 = model.train_step(*args, eps=eps, dropout_seed=11, apply_update=False)
        got = model.gradients()
        model.engine.comm_destroy()
    finally:
        model.engine.close()
    for k in got:                                                # deterministic mode: the exchange of one rank changes no bit
        assert np.array_equal(got[k], g_local[k]), k
    sc, flipped = kinks.compare(got, kinks.torch_oracle_run(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, 1.0, 11))
    assert abs(loss - sc["loss"]) < 1e-4 * max(1, abs(sc["loss"])) and len(flipped) <= 4


# ---- BASELINE config 3 at FULL size: T1 = train step B=32, T_text=128, T_mel=800 ---------------------------------------------
@pytest.mark.parametrize("rf", [2, 5])
def test_t1_full_size_train_step_properties(rf):
    """Too large for the float64 oracle in a test, so size-independent properties: finite losses, the gradient is a mean over
    utterances (a permuted batch gives the same gradient), per-utterance ELBO terms are permutation-equivariant, frozen
    statistics never receive a gradient, and -- against the fp32 autograd restatement on the host cores -- digests of a few
    variables spread over the model."""
    hps = LJHPS
    B, Tt, Tm = 32, 128, 800
    w = init_weights(hps, seed=1234, mode="synthetic")
    b = make_batch(B, Tt, Tm, ragged=True, seed=77, text_step=2, mel_step=11)
    r = np.random.Generator(np.random.PCG64(3))
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    Tz = (Tm + rf - 1) // rf
    eps = r.standard_normal((B, Tz, hps.Common.latent_dim)).astype(np.float32)
    perm = r.permutation(B)
    probe = ["decoder/pre_projection/kernel", "prior/glow/3/1/weight", "prior/glow/0/2/net/attentions/1/ffn/dense1/kernel",
             "posterior/attentions/0/att_proj2/kernel", "text_encoder/self_attentions/2/att_proj/kernel", "decoder/out_projection/bias",
             "prior/glow/5/0/log_scale", "posterior/pos_weight"]
    model = VAENAR(hps, weights=w)
    try:
        n0 = model.engine.launch_count()
        sc = model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1e-5, rf, eps=eps, dropout_seed=9, apply_update=False)
        launches = model.engine.launch_count() - n0
        g = model.gradients()                                   # every trainable variable (round 6: was 8 probes)
        assert all(np.isfinite(x) for x in sc) and launches < 3000
        # dropout OFF for the equivariance checks (the masks are indexed by row position): per-utterance ELBO terms, dev_step mode
        _, l2, kl, ll, _ = model(b["ids"], mels, b["mel_lengths"], b["text_lengths"], reduction_factor=rf, training=False, reduce_loss=False,
                                 eps=eps, return_alignments=False)
        l2, kl, ll = l2.numpy(), kl.numpy(), ll.numpy()
        _, l2p, klp, llp, _ = model(b["ids"][perm], mels[perm], b["mel_lengths"][perm], b["text_lengths"][perm], reduction_factor=rf,
                                    training=False, reduce_loss=False, eps=eps[perm], return_alignments=False)
        np.testing.assert_allclose(l2p.numpy(), l2[perm], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(llp.numpy(), ll[perm], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(klp.numpy(), kl[perm], rtol=2e-4, atol=2e-2)      # kl ~ 1e5 per utterance: sums of 51k terms
        assert np.isfinite(l2).all() and np.isfinite(kl).all() and np.isfinite(ll).all()
    finally:
        model.engine.close()
    # digests against the fp32 torch restatement (host cores; ~10-20 s)
    import torch
    from oracle.vaenar_torch import TorchOracle
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref, rsc = TorchOracle(hps, w, torch.float32, grad=True).gradients(b["ids"], mels, b["mel_lengths"], b["text_lengths"], rf, eps, kl_weight=1e-5,
                                                            length_weight=hps.Train.length_weight, dropout_seed=9)
    assert abs(sc[1] - rsc["mel_l2"]) < 1e-3 * max(1, abs(rsc["mel_l2"])) and abs(sc[3] - rsc["length_l2"]) < 1e-3 * max(1, abs(rsc["length_l2"]))
    assert abs(sc[2] - rsc["kl"]) < 2e-3 * max(1, abs(rsc["kl"]))
    # (200 M ReLU evaluations per step at this size: units on their kink flip on both sides, the fp32 restatement included; what that moves
    #  averages out in a kernel's largest entry but not in a SCALAR variable, whose gradient is one signed sum with cancellation: 3e-2 there)
    for k in probe:
        tol = 3e-2 if ref[k].size == 1 else 5e-3
        assert np.abs(g[k] - ref[k]).max() <= tol * np.abs(ref[k]).max() + 1e-7, (k, float(np.abs(g[k] - ref[k]).max()), float(np.abs(ref[k]).max()))
    # ALL variables, by relative 2-norm per tensor (VERDICT round 5 "next round" #3 (ii)): a unit on its ReLU kink moves single entries, not a
    # tensor's norm -- a wrong kernel, a missing term or a split defect moves the norm.  The reference is the fp32 restatement (its own
    # round-off and kink flips are in the budget): 2e-3 for tensors (1.0e-3 measured at worst), 3e-2 for scalar variables (one signed sum with cancellation, as above)
    assert set(g) == set(ref), sorted(set(g) ^ set(ref))[:5]
    # (a gradient that is ZERO in exact arithmetic -- the bias of a convolution that feeds a BatchNormalization on batch statistics without an
    #  activation in between, utils.py:76-85 with activation None: the mean is subtracted again -- is round-off on both sides; such a tensor is
    #  judged against the scale of the step's gradients, not against its own norm)
    gscale = max(float(np.abs(v).max()) for v in ref.values())
    rows = []
    for k in sorted(g):
        a, r_ = np.asarray(g[k], np.float64).ravel(), np.asarray(ref[k], np.float64).ravel()
        floor = 1e-6 * gscale * np.sqrt(a.size)
        rows.append((float(np.linalg.norm(a - r_)) / max(float(np.linalg.norm(r_)), floor), a.size, k, float(np.linalg.norm(r_)) < floor))
    rows.sort(reverse=True)
    print("T1 rf=%d: %d gradient tensors (largest |gradient| %.3g); worst relative 2-norm errors against the fp32 autograd restatement:" % (rf, len(rows), gscale))
    for e, n, k, z in rows[:8]:
        print("   %.3e  (%d elements)%s  %s" % (e, n, " [zero in exact arithmetic]" if z else "", k))
    print("   median %.3e" % float(np.median([e for e, _, _, _ in rows])))
    bad = [(e, n, k) for e, n, k, _ in rows if e > (3e-2 if n == 1 else 2e-3)]
    assert not bad, bad[:10]


# ---- integer frame counts: population test (inference.py:135-137, length_predictor.py:35-42) ---------------------------------
def test_frame_counts_population():
    """>= 2000 random ragged utterances: int32 frame counts of the engine (text encoder on the 3-term split path and on exact
    fp32 MFMA) against the float64 oracle.  A count may only differ where the float length itself sits on an integer
    boundary to within the fp32 noise of the sum (|frac - boundary| <= tol); the rates and the smallest margin are recorded."""
    from oracle.vaenar_numpy import Oracle
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    orc = Oracle(hps, w, np.float64)
    r = np.random.Generator(np.random.PCG64(2024))
    nb, B, T = 16, 128, 200
    pos_step = np.float32(hps.Common.mel_text_len_ratio) / np.float32(2)
    model = VAENAR(hps, weights=w)
    res = {"utterances": nb * B}
    try:
        lens_all, ref_all, got = [], [], {1: [], 0: []}
        for i in range(nb):
            tl = r.integers(8, T + 1, B).astype(np.int32)
            tl[0] = T
            ids = np.zeros((B, T), np.int32)
            for u in range(B):
                n = int(tl[u]); ids[u, 0] = 1; ids[u, 1:n - 1] = r.integers(3, 43, n - 2); ids[u, n - 1] = 2
            ref_all.append(orc.length_predictor(orc.text_encoder(ids, tl, pos_step=pos_step), tl))
            lens_all.append(tl)
            for split in (1, 0):
                model.engine.set_option("split_encoder", split)
                emb = model.text_encoder(ids, tl, pos_step=pos_step, training=False)
                got[split].append(model.length_predictor(emb, tl, training=False).numpy())
    finally:
        model.engine.close()
    ref = np.concatenate(ref_all)
    margin = np.minimum(ref - np.floor(ref), np.ceil(ref) - ref)
    res["min_boundary_margin_float64"] = float(margin.min())
    res["mean_frames"] = float(ref.mean())
    for split in (1, 0):
        g = np.concatenate(got[split]).astype(np.float64)
        mism = g.astype(np.int64) != ref.astype(np.int64)              # tf.cast(float, int32): truncation
        err = np.abs(g - ref)
        res["split_encoder=%d" % split] = {"mismatches": int(mism.sum()), "rate": float(mism.mean()), "max_abs_float_err": float(err.max()),
                                           "max_rel_float_err": float((err / ref).max()),
                                           "worst_margin_at_mismatch": float(margin[mism].max()) if mism.any() else None}
        # a differing count is only legitimate on an integer boundary: within the path's own float error of it
        assert (margin[mism] <= err.max() + 1e-12).all()
        assert (err / ref).max() < 2e-6
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "frame_counts_population.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("frame counts:", json.dumps(res))
    # the split path may not be worse than exact fp32 MFMA at deciding the integer
    assert res["split_encoder=1"]["mismatches"] <= res["split_encoder=0"]["mismatches"] + 1


# ---- n_sample > 1 in VAENAR.call (models.py:146-178) -----------------------------------------------------------------------------
def _nsample_fixture():
    with np.load(os.path.join(ROOT, "tests", "golden", "refshim_nsample2.npz")) as z:
        g = {k: z[k] for k in z.files}
    hps = tiny_hps()
    hps.Train.num_samples = int(g["n_sample"])
    w = init_weights(hps, seed=7, mode="synthetic")
    return g, hps, w


def test_n_sample_2_matches_reference_python():
    """hps.Train.num_samples = 2: decoded outputs for batch * n_sample latents, per-utterance terms averaged over the samples --
    against a fixture made by the reference's own Python (oracle/make_golden.py: build_ref_nsample)."""
    g, hps, w = _nsample_fixture()
    model = VAENAR(hps, weights=w)
    try:
        assert model.n_sample == 2
        outs, l2, kl, ll, ali = model(g["ids"], g["mels"], g["mel_lengths"], g["text_lengths"], reduction_factor=2, training=False,
                                      reduce_loss=False, eps=g["eps"])
        assert outs.shape == g["outs"].shape and np.abs(outs.numpy() - g["outs"]).max() < 2e-5
        np.testing.assert_allclose(l2.numpy(), g["l2"], rtol=1e-4)
        np.testing.assert_allclose(ll.numpy(), g["length"], rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(kl.numpy(), g["kl"], rtol=1e-3, atol=6e-2)
        for k in ali:
            np.testing.assert_allclose(ali[k].numpy(), g["ali_" + k], atol=1e-5)
        _, l2m, klm, llm, _ = model(g["ids"], g["mels"], g["mel_lengths"], g["text_lengths"], reduction_factor=2, training=False,
                                    reduce_loss=True, eps=g["eps"], return_alignments=False)
        np.testing.assert_allclose([l2m, llm], [g["l2"].mean(), g["length"].mean()], rtol=1e-3)
    finally:
        model.engine.close()


@pytest.mark.parametrize("B,Tt,Tm,text_step", [(3, 61, 150, 17), (2, 20, 70, 9), (2, 96, 66, 30), (3, 100, 130, 3), (2, 128, 90, 64)],
                         ids=["2-key-blocks", "1-block", "3-full-blocks", "4th-partial", "4-full-and-half"])
def test_blocks_without_alignments_fused_and_unfused(B, Tt, Tm, text_step):
    """attention.py:436-452 when nobody asks for the alignments (`return_alignments=False`: the decoder's blocks too): the chain
    launch that attends for its own 32 rows (engine option fuse_xattn) against the float64 oracle, LJ-sized model, ragged text and
    mel lengths, latent rows per utterance that are not a multiple of 32 (row panels straddle two utterances with different
    lengths), one to four key blocks with and without a partial last block, and the same call in the three-launch form; the fused
    form saves two launches per block."""
    from oracle.vaenar_numpy import Oracle
    hps = LJHPS
    w = init_weights(hps, seed=4321, mode="synthetic")
    b = make_batch(B, Tt, Tm, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True,
                   temperature=1.0, text_step=text_step, mel_step=31, seed=Tt)
    rmel, _ = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    model = VAENAR(hps, weights=w)
    try:
        launches, mels = {}, {}
        for fuse in (1, 0):
            model.engine.set_option("fuse_xattn", fuse)
            for rep in range(2):                      # (second call: every lazily built panel exists, the count is the steady state)
                n0 = model.engine.launch_count()
                mel, ali = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"],
                                           return_alignments=False)
                mels[fuse] = mel.numpy()
                launches[fuse] = model.engine.launch_count() - n0
            assert not ali
            assert np.abs(mels[fuse] - rmel).max() < 2e-5
        # (the fused launches run on the 4-wave chain kernel with panels that start at utterance boundaries, the three-launch form's chain C
        #  + coupling tail does not fit that kernel's LDS budget and runs on the 8-wave kernel: another accumulation order of the same
        #  split products)
        assert np.abs(mels[1] - mels[0]).max() < 2e-5
        assert launches[0] - launches[1] >= 2 * 12, launches        # 12 prior blocks + 2 decoder blocks, two launches saved each
    finally:
        model.engine.close()


def test_train_harness_on_tfrecords_with_test_synthesis(tmp_path):
    """F3 on the GPU: `{train,dev,test}-*.tfrecords` whose Examples were encoded by the protobuf LIBRARY
    (tests/tf_protos.py -- not this repository's codec) feed `train.py --data_dir` (/root/reference/train.py:81-110,
    datasets/tf_record_utils.py:108-142): one epoch with the init step, the dev pass and a checkpoint, then the periodic test
    synthesis of the reference's main loop (train.py:308-325: test_step -> TestUtils.synthesize_and_save_wavs): predicted mels and
    Griffin-Lim wavs of one test batch on disk."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_tf_formats import write_tfrecords_with_protobuf_library
    hps = tiny_hps()
    r = np.random.Generator(np.random.PCG64(12))
    items = []
    for i in range(14):
        n = 6 + i % 5
        text = np.concatenate([[1], r.integers(3, hps.Encoder.Transformer.vocab_size, n), [2]]).astype(np.int64)      # int64 text, float64 mels
        items.append(("utt%02d" % i, text, 0.5 * r.standard_normal((int(4.2 * len(text)) + i % 3, hps.Audio.num_mels))))
    rec_dir = tmp_path / "records"
    write_tfrecords_with_protobuf_library(str(rec_dir), {"train": items[:8], "dev": items[8:12], "test": items[12:]})
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--dataset", "tiny", "--data_dir", str(rec_dir), "--model_dir", str(tmp_path / "model"),
           "--log_dir", str(tmp_path / "log"), "--test_dir", str(tmp_path / "test"), "--test_interval", "1", "--epochs", "1", "--batch_size", "4",
           "--seed", "5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2000:])
    so = out.stdout
    assert "Initial step" in so and "Training Epoch 1," in so and "dev-total" in so and "Saved checkpoint for epoch" in so
    steps = [ln for ln in so.splitlines() if ln.startswith("Step ")]
    assert len(steps) == 2                                       # 8 training utterances in global batches of 4
    vals = [float(x) for ln in steps for x in __import__("re").findall(r"-?\d+\.\d+", ln.rsplit(", time", 1)[0])]
    assert np.isfinite(vals).all()
    assert "Testing ..." in so and "All wavs for test are synthesized!" in so and "test finished" in so
    made = sorted(os.listdir(tmp_path / "test"))
    assert [f for f in made if f.endswith(".wav")] == ["test-utt12-1.wav", "test-utt13-1.wav"], made
    assert [f for f in made if f.endswith(".npy")] == ["test-utt12-1.npy", "test-utt13-1.npy"], made
    mel = np.load(tmp_path / "test" / "test-utt12-1.npy")
    assert mel.shape == (items[12][2].shape[0], hps.Audio.num_mels) and np.isfinite(mel).all()
    assert os.path.getsize(tmp_path / "test" / "test-utt12-1.wav") > 1000


def test_kv_overlap_option_gives_the_same_bits():
    """Engine option kv_overlap (round 6 experiment, default off: measured 1 % slower): the prior's cross K | V projection on a second
    stream beside the first flow step's pre-chain and self-attention.  Same kernels, same operands, another schedule: bit-identical mels
    and alignments, also with several calls in flight behind each other."""
    hps = LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    b = make_batch(4, 37, 150, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, temperature=1.0,
                   text_step=5, mel_step=23)
    model = VAENAR(hps, weights=w)
    try:
        mel0, ali0 = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
        ref, refa = mel0.numpy(), {k: v.numpy() for k, v in ali0.items()}
        model.engine.set_option("kv_overlap", 1)
        outs = [model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"]) for _ in range(3)]
        for mel, ali in outs:
            assert np.array_equal(mel.numpy(), ref)
            for k in refa:
                assert np.array_equal(ali[k].numpy(), refa[k])
    finally:
        model.engine.close()
