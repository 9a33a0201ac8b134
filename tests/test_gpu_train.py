"""Training step on the GPU (vnr_train_step) against the autograd restatement (oracle/vaenar_torch.py, float64):
losses, the gradient of EVERY trainable variable, and the Adam update (train.py:127-138)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import kinks
from oracle.vaenar_torch import TorchOracle, adam_step
from vaenar_tts_amd.configs import LJHPS, tiny_hps
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights, is_trainable

pytestmark = pytest.mark.gpu


def _case(name):
    hps = tiny_hps() if name.startswith("tiny") else LJHPS
    w = init_weights(hps, seed=1234, mode="synthetic")
    if name == "tiny-long":      # T_z = 550 latent frames: probabilities beyond 512 keys (two-pass form), 18 query tiles
        b = make_batch(2, 23, 1100, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                       ragged=True, text_step=5, mel_step=211)
    elif name == "tiny-mid":     # T_z = 132, T_text = 12 (multiples of 4: the third-form attention backward on self- AND cross-attention; five
        b = make_batch(2, 12, 264, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,   # query tiles, so the
                       ragged=True, text_step=3, mel_step=62)                                                           # two-query-group dK / dV form runs)
    elif name == "lj-mid":       # LJ widths, 264 latent rows: the third-generation kernel-gradient GEMM (M >= 256, K and N >= 128) runs
        b = make_batch(2, 12, 264, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                       ragged=True, text_step=3, mel_step=62)
    elif name == "tiny":
        b = make_batch(3, 11, 40, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                       ragged=True, text_step=3, mel_step=7)
    else:
        b = make_batch(2, 19, 46, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                       ragged=True, text_step=5, mel_step=9)
    r = np.random.Generator(np.random.PCG64(31))
    B, Tm = len(b["mel_lengths"]), int(b["mel_lengths"].max())
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((B, (Tm + 1) // 2, hps.Common.latent_dim)).astype(np.float32)
    return hps, w, b, mels, eps


def _rel_err(a, b):
    scale = max(np.abs(b).max(), 1e-12)
    return float(np.abs(a - b).max() / scale)


@pytest.mark.parametrize("name,kw,recompute,chain", [("tiny", 1.0, 0, 1), ("tiny", 1e-5, 0, 1), ("lj", 1.0, 0, 1), ("tiny", 1.0, 1, 1), ("lj", 1.0, 1, 1),
                                                        ("tiny-long", 1.0, 0, 1), ("tiny-long", 1.0, 1, 1), ("tiny-mid", 1.0, 0, 1),
                                                        ("lj", 1.0, 0, 0), ("lj", 1.0, 0, 2), ("lj", 1e-5, 0, 3), ("lj", 1.0, 0, 11), ("lj", 1e-5, 0, 12)])
def test_gradients_match_autograd(name, kw, recompute, chain):
    """kl_weight = 1 makes the flow / posterior-entropy terms as visible as the L2 terms (the schedule value 1e-5 of
    train.py:236-243 is covered too).  recompute = 1: engine option "attn_bwd_recompute" -- the attention backward rebuilds the
    probabilities from Q, K and the forward's row statistics instead of reading stored ones (ragged lengths, causal and cross).
    chain: engine option "train_chain" -- the forward of every CrossAttentionBLK as two chain launches (1 = default, LJ-sized
    blocks only; 2 / 3 = 64- / 32-row panels forced) or as separate GEMM / LayerNorm launches (0); by default the backward of those
    blocks runs as two backward-chain launches as well ("train_chain_bwd", gemm3b.hip; 11 / 12: forward chains, unfused backward)."""
    hps, w, b, mels, eps = _case(name)
    model = VAENAR(hps, weights=w)
    model.engine.set_option("attn_bwd_recompute", recompute)
    model.engine.set_option("train_chain", chain % 10)          # (1x: forward chains with the unfused backward)
    model.engine.set_option("train_chain_bwd", 0 if chain >= 10 else 1)
    try:
        loss, mel_l2, kl, len_l2 = model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], kw, 2, eps=eps,
                                                    dropout_seed=11, apply_update=False)
        got = model.gradients()
    finally:
        model.engine.close()
    # every gradient at 2e-3 of the tensor's largest entry (+ an absolute floor: e.g. the bias of the last PostNet convolution sits
    # directly in front of a BatchNormalization, its true gradient is exactly 0); hidden units within float32 rounding of their ReLU
    # kink are identified and the oracle re-run with the engine's side of the kink (oracle/kinks.py)
    sc, flipped = kinks.compare(got, kinks.torch_oracle_run(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, kw, 11))
    assert abs(mel_l2 - sc["mel_l2"]) < 2e-5 * max(1, abs(sc["mel_l2"]))
    assert abs(len_l2 - sc["length_l2"]) < 1e-4 * max(1, abs(sc["length_l2"]))
    assert abs(kl - sc["kl"]) < 1e-4 * max(1, abs(sc["kl"]))
    assert abs(loss - sc["loss"]) < 1e-4 * max(1, abs(sc["loss"]))
    assert len(flipped) <= 3, flipped


def test_adam_update_matches_keras_formula():
    """Three optimizer steps.  The update of every variable must equal Keras Adam (train.py:116-117) applied to the
    engine's own gradients (Adam divides by sqrt(v), so entries whose gradient is rounding noise cannot be compared
    across two different gradient computations); the gradients themselves are pinned by the test above, and once more
    here at the weights reached after two updates."""
    hps, w, b, mels, eps = _case("tiny")
    model = VAENAR(hps, weights=w)
    # hps.Train.learning_rate (1e-3, the reference's).  On this trajectory the weights reached after one update put ONE hidden unit of
    # one FFN within float32 rounding of zero: its ReLU mask is not defined at fp32 resolution (it follows the last bits of the
    # forward pass) and one flipped mask is a 4e-3 error in that FFN's kernel gradients (tools/archive/r03_adam_err.py).  oracle/kinks.py finds
    # such units and compares against the oracle run on the engine's side of the kink, at the usual 2e-3.
    LR = 1e-3
    try:
        m = v = None
        for step in (1, 2, 3):
            before = model.get_weights()
            model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2, eps=eps, dropout_seed=step,
                             learning_rate=LR, apply_update=True)
            g = {k: x.astype(np.float64) for k, x in model.gradients().items()}
            after = model.get_weights()
            if m is None:
                m = {k: np.zeros_like(x) for k, x in g.items()}; v = {k: np.zeros_like(x) for k, x in g.items()}
            ref = {k: before[k].astype(np.float64) for k in g}
            adam_step(ref, g, m, v, step, lr=LR)
            for k in sorted(g):
                np.testing.assert_allclose(after[k], ref[k], rtol=2e-6, atol=2e-7, err_msg="%s step %d" % (k, step))
            if step == 2:      # gradients at the updated weights (transposed kernels, inverse flow matrices, scalars refreshed)
                _, flipped = kinks.compare(g, kinks.torch_oracle_run(hps, before, b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, 1.0, step))
                assert len(flipped) <= 3, flipped
        # BN moving statistics are assigned by the forward, not optimised: three momentum-0.99 updates moved them
        k = "decoder/postnet/conv_stack/0/bn/moving_mean"
        assert np.abs(after[k] - w[k]).max() > 1e-4
        # inference after training steps runs on the lazily re-packed panels
        mel, _ = model.inference(b["ids"], b["mel_lengths"], b["text_lengths"], eps=b["eps"])
        assert np.isfinite(mel.numpy()).all()
    finally:
        model.engine.close()


def test_weights_uploaded_between_steps_reach_the_next_step():
    """The transposed / flipped / split copies of the kernels are derived behind an optimizer step (not at the top of the next one);
    an upload in between (a checkpoint restore: vnr_set_weight) must invalidate them.  Step, restore the initial weights, then the
    gradients must be those of a fresh engine at the initial weights -- bit for bit apart from the atomic accumulation order."""
    hps, w, b, mels, eps = _case("tiny")
    args = (b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2)
    fresh = VAENAR(hps, weights=w)
    try:
        fresh.train_step(*args, eps=eps, dropout_seed=5, apply_update=False)
        g0 = fresh.gradients()
    finally:
        fresh.engine.close()
    model = VAENAR(hps, weights=w)
    try:
        model.train_step(*args, eps=eps, dropout_seed=4, learning_rate=1e-2, apply_update=True)      # moves every kernel by ~1e-2
        model.load_weights(w)
        model.train_step(*args, eps=eps, dropout_seed=5, apply_update=False)
        g1 = model.gradients()
    finally:
        model.engine.close()
    for k in sorted(g0):
        scale = max(np.abs(g0[k]).max(), 1e-12)
        assert np.abs(g1[k] - g0[k]).max() <= 1e-4 * scale + 1e-9, k


def test_rccl_allreduce_path_single_rank():
    """The gradient exchange of data-parallel training (RCCL all-reduce of the flat gradient between backward and Adam)
    on a one-rank communicator: the call sequence runs on the device and leaves the gradients unchanged (sum over one
    rank, times 1/1).  Multi-rank sharding logic is covered on CPU by tests/test_dist_gloo.py."""
    hps, w, b, mels, eps = _case("tiny")
    model = VAENAR(hps, weights=w)
    try:
        args = (b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2)
        model.train_step(*args, eps=eps, dropout_seed=3, apply_update=False)
        g0 = model.gradients()
        model.engine.comm_init(1, 0, model.engine.comm_unique_id())
        model.engine.comm_broadcast_weights()
        model.train_step(*args, eps=eps, dropout_seed=3, apply_update=False)
        g1 = model.gradients()
        model.engine.comm_destroy()
    finally:
        model.engine.close()
    for k in g0:
        assert np.abs(g1[k] - g0[k]).max() <= 1e-5 * np.abs(g0[k]).max() + 1e-7, k    # (float atomics: summation order varies)


def test_deterministic_training_mode():
    """Engine option "deterministic" (the reference pins TF_DETERMINISTIC_OPS=1 and every seed, train.py:17-32): the kernel-gradient
    GEMMs' row splits, the column sums, the LayerNorm / BatchNorm / ActNorm / embedding / length-predictor gradients leave ordered
    partials instead of float atomics.  Two identical steps then give bit-identical gradients, three identical runs of two optimizer
    steps bit-identical variables; the gradients agree with the default (atomic) mode to rounding."""
    hps, w, b, mels, eps = _case("lj-mid")
    args = (b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2)

    def run(det, steps):
        model = VAENAR(hps, weights=w)
        try:
            model.engine.set_option("deterministic", det)
            out = []
            for i in range(steps):
                model.train_step(*args, eps=eps, dropout_seed=7 + i, apply_update=steps > 1)
                out.append(model.gradients())
            return out, (model.get_weights() if steps > 1 else None)
        finally:
            model.engine.close()

    (g0,), _ = run(1, 1)
    (g1,), _ = run(1, 1)
    for k in sorted(g0):
        assert np.array_equal(g0[k], g1[k]), "gradient of %s differs between two identical deterministic steps" % k
    (ga,), _ = run(0, 1)
    for k in sorted(g0):
        assert np.abs(ga[k] - g0[k]).max() <= 1e-4 * max(np.abs(g0[k]).max(), 1e-12) + 1e-9, k
    runs = [run(1, 2) for _ in range(3)]
    for gs, ws in runs[1:]:
        for k in sorted(ws):
            assert np.array_equal(ws[k], runs[0][1][k]), "variable %s differs after two deterministic optimizer steps" % k
        for k in sorted(gs[1]):
            assert np.array_equal(gs[1][k], runs[0][0][1][k]), k


_FORK_CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import numpy as np
from vaenar_tts_amd.configs import tiny_hps
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
hps = tiny_hps()
w = init_weights(hps, seed=31, mode="synthetic")
b = make_batch(3, 13, 44, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, seed=3, text_step=3, mel_step=7)
r = np.random.Generator(np.random.PCG64(4))
mels = r.standard_normal((3, 44, hps.Audio.num_mels)).astype(np.float32)
eps = r.standard_normal((3, 22, hps.Common.latent_dim)).astype(np.float32)
m = VAENAR(hps, weights=w)
m.engine.set_option("deterministic", 1)
out = []
for rep in range(3):
    m.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2, eps=eps, dropout_seed=5, apply_update=False)
    g = m.gradients()
    out.append({k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16] for k, v in g.items()})
enc = {k: float(np.abs(v).max()) for k, v in g.items() if k.startswith("text_encoder/")}
print(json.dumps({"digests": out, "enc_max": enc}))
m.engine.close()
"""


def _fork_child(env_extra):
    import json
    import subprocess
    import sys
    env = dict(os.environ); env.update(env_extra)
    cp = subprocess.run([sys.executable, "-c", _FORK_CHILD % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert cp.returncode == 0, cp.stderr[-2000:]
    return json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])


def test_decoder_branch_stays_on_the_main_stream_without_the_stacked_projection():
    """the decoder branch may leave the main stream only while the stacked cross K | V projection is active.
    Without it (VNR_TRAIN_NO_KV_STACK=1) every decoder block adds its own data gradient into the text encoding's gradient; forked, that
    raced with the prior's blocks and with the arena clear -- text-encoder gradients wrong and irreproducible, silently.  Child
    processes (the switches are read once per process): deterministic mode, three identical steps each -- the un-stacked run must be
    reproducible and agree bit for bit with the one-stream schedule."""
    a = _fork_child({"VNR_TRAIN_NO_KV_STACK": "1"})
    b = _fork_child({"VNR_TRAIN_NO_KV_STACK": "1", "VNR_TRAIN_ONE_STREAM": "1"})
    assert a["digests"][0] == a["digests"][1] == a["digests"][2]
    assert a["digests"][0] == b["digests"][0]
    assert all(v > 0 for v in a["enc_max"].values())


_GROUP_CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
hps = LJHPS
w = init_weights(hps, seed=1234, mode="synthetic")
b = make_batch(2, 12, 528, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, text_step=3, mel_step=62)
r = np.random.Generator(np.random.PCG64(31))
Tm = int(b["mel_lengths"].max())
mels = r.standard_normal((2, Tm, hps.Audio.num_mels)).astype(np.float32)
eps = r.standard_normal((2, (Tm + 1) // 2, hps.Common.latent_dim)).astype(np.float32)
m = VAENAR(hps, weights=w)
m.engine.set_option("deterministic", 1)
n0 = m.engine.launch_count()
sc = m.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2, eps=eps, dropout_seed=5, apply_update=False)
launches = m.engine.launch_count() - n0
g = m.gradients()
print(json.dumps({"digests": {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16] for k, v in g.items()}, "launches": launches,
                  "scalars": [float(x) for x in sc]}))
m.engine.close()
"""


def test_grouped_kernel_gradient_launches_keep_every_bit():
    """Round 6: the kernel-gradient GEMMs dW = X^T.dY (one per dense / convolution tap, /root/reference/train.py:127-138 under the tape) leave the
    training step three at a time as ONE launch (csrc/train_kernels.hip: gemm_tn3_group_kernel; csrc/train.inc: tn_enqueue / tn_flush) -- every job
    with the tiles, row splits and ordered partials its own launch would have had.  Deterministic mode, LJ widths, 528 latent rows (the
    third-generation kernel takes M >= 256): every gradient of one job per launch (VNR_GEMM_TN_GROUP=1, rounds 2-5), of the default and of
    eight per launch is the same, bit for bit -- and the grouped runs really are grouped (fewer launches)."""
    import json
    import subprocess
    import sys

    def child(group):
        env = dict(os.environ)
        if group: env["VNR_GEMM_TN_GROUP"] = str(group)
        else: env.pop("VNR_GEMM_TN_GROUP", None)
        cp = subprocess.run([sys.executable, "-c", _GROUP_CHILD % ROOT], capture_output=True, text=True, env=env, timeout=900)
        assert cp.returncode == 0, cp.stderr[-2000:]
        return json.loads([ln for ln in cp.stdout.splitlines() if ln.startswith("{")][-1])
    one, dflt, eight = child(1), child(0), child(8)
    assert one["digests"] == dflt["digests"] == eight["digests"]
    assert one["scalars"] == dflt["scalars"] == eight["scalars"]
    assert eight["launches"] < dflt["launches"] < one["launches"], (one["launches"], dflt["launches"], eight["launches"])
    assert one["launches"] - dflt["launches"] >= 100          # (292 jobs: 3 per launch saves ~190 launches)


@pytest.mark.parametrize("name,kw,det", [("tiny", 1.0, 0), ("lj", 1.0, 0), ("lj", 1e-5, 1), ("tiny-mid", 1.0, 1)])
def test_train_step_through_inverse_flows(name, kw, det):
    """Prior.Transformer.inverse = True (/root/reference/modules/prior.py:81,88-99; no shipped configuration sets it): every flow of the prior
    is built with the flag and BaseFlow.bwd_pass -- what TransformerPrior.log_probability calls, prior.py:119-152 -- runs the _FORWARD passes
    (/root/reference/modules/flow.py:91-113: coupling 223-239, InvertibleLinear 123-135, ActNorm 166-175).  The training step follows (round 6;
    it refused the option by name before): scalars and ALL gradients against the autograd restatement with the same flag, whose forward is
    pinned to the NumPy specification and through it to the reference's own flow code (tests/test_oracle_torch.py, tests/golden/refshim_inverse.npz).
    det = 1: the deterministic mode (ordered sums instead of float atomics) runs the same closures."""
    import copy
    hps, w, b, mels, eps = _case(name)
    hps = copy.deepcopy(hps)                             # (LJHPS is a module-level object)
    hps.Prior.Transformer.inverse = True
    model = VAENAR(hps, weights=w)
    try:
        model.engine.set_option("deterministic", det)
        loss, mel_l2, kl, len_l2 = model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], kw, 2, eps=eps,
                                                    dropout_seed=11, apply_update=False)
        got = model.gradients()
        if det:                                          # bit-reproducible
            model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], kw, 2, eps=eps, dropout_seed=11, apply_update=False)
            again = model.gradients()
            assert all(np.array_equal(got[k], again[k]) for k in got)
    finally:
        model.engine.close()
    sc, flipped = kinks.compare(got, kinks.torch_oracle_run(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, kw, 11))
    assert abs(mel_l2 - sc["mel_l2"]) < 2e-5 * max(1, abs(sc["mel_l2"]))
    assert abs(len_l2 - sc["length_l2"]) < 1e-4 * max(1, abs(sc["length_l2"]))
    assert abs(kl - sc["kl"]) < 1e-4 * max(1, abs(sc["kl"]))
    assert abs(loss - sc["loss"]) < 1e-4 * max(1, abs(sc["loss"]))
    assert len(flipped) <= 3, flipped
    # the flag changed something: the same step without it has another KL term
    hps2, w2, b2, mels2, eps2 = _case(name)
    plain = VAENAR(hps2, weights=w2)
    try:
        _, _, kl0, _ = plain.train_step(b2["ids"], mels2, b2["text_lengths"], b2["mel_lengths"], kw, 2, eps=eps2, dropout_seed=11, apply_update=False)
    finally:
        plain.engine.close()
    assert abs(kl0 - kl) > 1e-3 * max(1.0, abs(kl))
