#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (test infrastructure).

    python oracle/make_golden.py            # oracle fixtures + (when /root/reference exists) refshim_* fixtures

Each fixture holds inputs and expected outputs of VAENAR.inference and of inference.py's test_step
on a small ragged batch; weights are NOT stored -- they are regenerated from (config, seed, mode) by
vaenar_tts_amd.weights.init_weights and pinned by a digest.  The reference itself cannot produce
vectors (it imports TensorFlow 2.2, absent here): PARITY UNPINNED, see oracle/vaenar_numpy.py.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.vaenar_numpy import Oracle  # noqa: E402
from vaenar_tts_amd.configs import DataBakerHPS, LJHPS, tiny_hps  # noqa: E402
from vaenar_tts_amd.synthetic import make_batch  # noqa: E402
from vaenar_tts_amd.weights import init_weights  # noqa: E402

CASES = {
    # name: (hps factory, batch kwargs)
    "tiny_ragged": (tiny_hps, dict(B=3, T_text=11, T_mel=40, ragged=True, temperature=1.0, text_step=3, mel_step=7)),
    "lj_ragged": (lambda: LJHPS, dict(B=4, T_text=37, T_mel=150, ragged=True, temperature=1.0, text_step=5, mel_step=23)),
    "lj_t0": (lambda: LJHPS, dict(B=2, T_text=21, T_mel=64, ragged=True, temperature=0.0, text_step=6, mel_step=14)),
}
SEED = 1234


def weights_digest(w):
    h = hashlib.sha256()
    for k in w:
        h.update(k.encode()); h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


def build(name):
    mk, kw = CASES[name]
    hps = mk()
    w = init_weights(hps, seed=SEED, mode="synthetic")
    o = Oracle(hps, w, np.float64)
    b = make_batch(vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, seed=SEED, **kw)
    mel, ali = o.inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
    out = dict(ids=b["ids"], text_lengths=b["text_lengths"], mel_lengths=b["mel_lengths"], eps=b["eps"],
               text_embd=o.last["text_embd"], z=o.last["z"], prior_logprobs=o.last["prior_logprobs"],
               initial=o.last["initial"], mel=mel)
    for k, v in ali.items():
        out["ali_" + k] = v
    tmel, tlen, _ = o.test_step(b["ids"], b["text_lengths"])
    out.update(ts_pred_float=o.last["pred_float"], ts_lengths=tlen, ts_mel=tmel)
    out = {k: (v.astype(np.float32) if v.dtype == np.float64 and k != "ts_pred_float" else v) for k, v in out.items()}
    out["weights_sha256"] = np.frombuffer(weights_digest(w).encode(), dtype=np.uint8)
    return out


# Fixtures produced by the REFERENCE's own Python (models.VAENAR / modules.* imported from /root/reference and
# executed over oracle/tf_shim in float64).  They pin the composition (call order, concat order, masks, head
# swap ...) of the path; TensorFlow's kernel numerics stay unpinned.
REF_CASES = {
    "refshim_tiny": (tiny_hps, dict(B=3, T_text=11, T_mel=40, ragged=True, temperature=1.0, text_step=3, mel_step=7), 7),
    "refshim_lj": (lambda: LJHPS, dict(B=2, T_text=19, T_mel=50, ragged=True, temperature=1.0, text_step=6, mel_step=13), 11),
    # BASELINE config 5's model: the reference's DataBakerHPS (hparams.py:351-474: vocabulary 39, mel / text length ratio 4.21)
    "refshim_databaker": (lambda: DataBakerHPS, dict(B=2, T_text=17, T_mel=46, ragged=True, temperature=1.0, text_step=5, mel_step=11), 13),
}


def build_ref(name):
    from oracle.run_reference_on_shim import reference_call, reference_inference
    mk, kw, seed = REF_CASES[name]
    hps = mk()
    w = init_weights(hps, seed=seed, mode="synthetic")
    b = make_batch(vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, seed=seed, **kw)
    mel, ali, _, _ = reference_inference(hps, w, b["ids"], b["mel_lengths"], b["text_lengths"], b["eps"])
    r = np.random.Generator(np.random.PCG64(seed + 1))
    Tm = int(b["mel_lengths"].max())
    mels = r.standard_normal((len(b["mel_lengths"]), Tm, hps.Audio.num_mels)).astype(np.float32)
    eps4 = r.standard_normal((len(b["mel_lengths"]), 1, (Tm + 1) // 2, hps.Common.latent_dim)).astype(np.float32)
    outs, l2, kl, ll, cali = reference_call(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"],
                                            eps4.reshape(eps4.shape[0], 1, eps4.shape[2], eps4.shape[3]))
    out = dict(seed=np.int64(seed), ids=b["ids"], text_lengths=b["text_lengths"], mel_lengths=b["mel_lengths"], eps=b["eps"],
               mel=np.asarray(mel, np.float32), call_mels=mels, call_eps=eps4, call_outs=np.asarray(outs, np.float32),
               call_l2=np.asarray(l2, np.float64), call_kl=np.asarray(kl, np.float64), call_length=np.asarray(ll, np.float64))
    for k, v in ali.items():
        out["ali_" + k] = np.asarray(v, np.float32)
    out["weights_sha256"] = np.frombuffer(weights_digest(w).encode(), dtype=np.uint8)
    return out


def build_ref_nsample(n_sample=2):
    """VAENAR.call with hps.Train.num_samples = 2 (models.py:146-178), evaluation mode, by the reference's own Python."""
    from oracle.run_reference_on_shim import reference_call
    hps = tiny_hps()
    hps.Train.num_samples = n_sample
    w = init_weights(hps, seed=7, mode="synthetic")
    b = make_batch(vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, seed=7, B=3, T_text=11, T_mel=40,
                   ragged=True, text_step=3, mel_step=7)
    r = np.random.Generator(np.random.PCG64(8))
    mels = r.standard_normal((3, 40, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((3, n_sample, 20, hps.Common.latent_dim)).astype(np.float32)
    outs, l2, kl, ll, ali = reference_call(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], eps)
    out = dict(n_sample=np.int64(n_sample), ids=b["ids"], text_lengths=b["text_lengths"], mel_lengths=b["mel_lengths"], mels=mels, eps=eps,
               outs=np.asarray(outs, np.float32), l2=np.asarray(l2, np.float64), kl=np.asarray(kl, np.float64), length=np.asarray(ll, np.float64),
               weights_sha256=np.frombuffer(weights_digest(w).encode(), np.uint8))
    for k, v in ali.items():
        out["ali_" + k] = np.asarray(v, np.float32)
    return out


def build_train(name="train_tiny"):
    """Training-step fixture (train.py:127-138) from the autograd restatement: losses, the gradient of every trainable
    variable and the displacement of every variable after one Keras Adam step, each as a 19-number digest."""
    from oracle.vaenar_torch import TorchOracle, adam_step
    from vaenar_tts_amd.configs import tiny_hps
    from vaenar_tts_amd.synthetic import make_batch
    from vaenar_tts_amd.weights import init_weights, is_trainable
    hps = tiny_hps()
    w = init_weights(hps, seed=SEED, mode="synthetic")
    b = make_batch(3, 11, 40, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim,
                   ragged=True, text_step=3, mel_step=7)
    r = np.random.Generator(np.random.PCG64(31))
    mels = r.standard_normal((3, 40, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((3, 20, hps.Common.latent_dim)).astype(np.float32)
    out = dict(ids=b["ids"], text_lengths=b["text_lengths"], mel_lengths=b["mel_lengths"], mels=mels, eps=eps,
               kl_weight=np.float64(1.0), dropout_seed=np.int64(11), reduction_factor=np.int64(2),
               weights_sha256=np.frombuffer(weights_digest(w).encode(), np.uint8))
    g, sc = TorchOracle(hps, w).gradients(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, kl_weight=1.0,
                                          length_weight=hps.Train.length_weight, dropout_seed=11)
    out["scalars"] = np.array([sc["loss"], sc["mel_l2"], sc["kl"], sc["length_l2"]], np.float64)
    ww = {k: np.asarray(v, np.float64).copy() for k, v in w.items()}
    m = {k: np.zeros_like(v) for k, v in ww.items() if is_trainable(k)}; v = {k: np.zeros_like(x) for k, x in m.items()}
    adam_step(ww, g, m, v, 1, lr=hps.Train.learning_rate)
    for k in sorted(g):                    # digests keep the fixture small: 16 strided samples + sum + l2 norm + max |.|
        out["grad/" + k] = digest(g[k])
        out["adam1/" + k] = digest(ww[k] - np.asarray(w[k], np.float64))
    return out


# Fixtures of the TRAINING path produced by the REFERENCE's own Python (models.VAENAR / modules.* imported from /root/reference and
# executed over oracle/tf_shim_torch: float64 torch tensors, torch.autograd in place of tf.GradientTape): VAENAR.call with
# training=True (Dropout on the engine's counter-based masks, BatchNormalization on batch statistics), the loss of train.py:135
# and d loss / d every trainable variable, the moving statistics after the forward, and VAENAR.init (models.py:212-226).
REF_TRAIN_CASES = {
    # name: (hps factory, B, T_text, T_mel, text_step, mel_step, rf, weight seed)
    "refshim_train_tiny": (tiny_hps, 3, 11, 40, 3, 7, 2, SEED),
    "refshim_train_lj": (lambda: LJHPS, 2, 19, 46, 5, 9, 2, SEED),
    # Prior.Transformer.inverse = True (prior.py:81,88-99): log_probability then runs the flows' _forward passes (flow.py:91-113) and
    # init / sample the _backward ones -- the training step through them, by the reference's own Python (round 6)
    "refshim_train_tiny_inv": (lambda: _inverse(tiny_hps()), 3, 11, 40, 3, 7, 2, SEED),
}


def _inverse(hps):
    hps.Prior.Transformer.inverse = True
    return hps


def build_ref_train(name):
    from oracle.run_reference_on_shim import reference_call_training, reference_init, reference_train_step
    mk, B, Tt, Tm, ts, ms, rf, seed = REF_TRAIN_CASES[name]
    hps = mk()
    w = init_weights(hps, seed=seed, mode="synthetic")
    b = make_batch(B, Tt, Tm, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True,
                   text_step=ts, mel_step=ms)
    r = np.random.Generator(np.random.PCG64(31))
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    eps = r.standard_normal((B, (Tm + rf - 1) // rf, hps.Common.latent_dim)).astype(np.float32)
    out = dict(ids=b["ids"], text_lengths=b["text_lengths"], mel_lengths=b["mel_lengths"], mels=mels, eps=eps,
               reduction_factor=np.int64(rf), dropout_seed=np.int64(11), weight_seed=np.int64(seed),
               weights_sha256=np.frombuffer(weights_digest(w).encode(), np.uint8))
    for tag, kw in (("kw1", 1.0), ("kw1e-5", 1e-5)):           # kl weight 1 (all terms visible) and the schedule's 1e-5
        sc, g, preds, stats = reference_train_step(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], eps, rf, kw, 11)
        out[tag + "/kl_weight"] = np.float64(kw)
        out[tag + "/scalars"] = np.array([sc["loss"], sc["mel_l2"], sc["kl"], sc["length_l2"]], np.float64)
        for k in sorted(g):
            out[tag + "/gdig/" + k] = digest(g[k])
            if tag == "kw1" and g[k].size <= 4096:              # small variables in full (biases, LayerNorm / ActNorm / scalars ...)
                out["kw1/grad/" + k] = g[k].astype(np.float32)
        if tag == "kw1":
            out["predictions"] = preds.astype(np.float32)
            for k, v in stats.items():
                out["moving/" + k] = v.astype(np.float64)
    outs, l2, kl, ll, ali = reference_call_training(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], eps, rf, 11)
    out.update(call_l2=l2.astype(np.float64), call_kl=kl.astype(np.float64), call_length=ll.astype(np.float64))
    for k, v in ali.items():
        out["call_ali/" + k] = v.astype(np.float32)
    # VAENAR.init at max_reduction_factor (train.py:176-179, 262)
    mrf = hps.Common.max_reduction_factor
    Tz5 = int(((b["mel_lengths"].astype(np.int64) + mrf - 1) // mrf).max())
    eps5 = r.standard_normal((B, Tz5, hps.Common.latent_dim)).astype(np.float32)
    mel, after = reference_init(hps, w, b["ids"], b["mel_lengths"], b["text_lengths"], eps5, 5)
    out.update(init_eps=eps5, init_dropout_seed=np.int64(5), init_mel=mel.astype(np.float32))
    for k, v in after.items():
        if np.abs(v - np.asarray(w[k], np.float64)).max() > 0:          # what init changed: ActNorm variables, BN moving statistics
            out["init/" + k] = v.astype(np.float64)
    return out


def build_ref_b2():
    """tests/golden/refshim_b2.npz -- the rest of the call surface models.VAENAR uses, by the REFERENCE's
    own Python over the torch shim (tiny configuration): (i) the module-level posterior / prior methods
    (run_reference_on_shim.reference_module_methods); (ii) train_step with hps.Train.num_samples = 2 (models.py:141-178 under
    training=True and autograd): scalars, predictions, the gradient of every trainable variable."""
    from oracle.run_reference_on_shim import reference_call_training, reference_module_methods, reference_train_step
    hps = tiny_hps()
    w = init_weights(hps, seed=SEED, mode="synthetic")
    B, Tt, Tm, rf, ns = 3, 11, 40, 2, 2
    b = make_batch(B, Tt, Tm, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True,
                   text_step=3, mel_step=7)
    r = np.random.Generator(np.random.PCG64(41))
    C, Tz = hps.Common.latent_dim, (Tm + rf - 1) // rf
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    zl = ((b["mel_lengths"].astype(np.int64) + rf - 1) // rf).astype(np.int32)
    eps_post = r.standard_normal((B, ns, Tz, C)).astype(np.float32)
    eps_prior = r.standard_normal((B, int(zl.max()), C)).astype(np.float32)
    eps_init = r.standard_normal((B, int(zl.max()), C)).astype(np.float32)
    out = dict(ids=b["ids"], text_lengths=b["text_lengths"], mel_lengths=b["mel_lengths"], z_lengths=zl, mels=mels, eps_post=eps_post,
               eps_prior=eps_prior, eps_init=eps_init, n_sample=np.int64(ns), reduction_factor=np.int64(rf), dropout_seed=np.int64(11),
               weights_sha256=np.frombuffer(weights_digest(w).encode(), np.uint8))
    mm = reference_module_methods(hps, w, b["ids"], b["text_lengths"], mels[:, ::rf], zl, eps_post, eps_prior, eps_init, 11)
    for k, v in mm.items():
        out["mod/" + k] = np.asarray(v, np.float64)
    hps.Train.num_samples = ns
    for tag, kw in (("kw1", 1.0), ("kw1e-5", 1e-5)):
        sc, g, preds, stats = reference_train_step(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], eps_post, rf, kw, 11)
        out["ns2/" + tag + "/scalars"] = np.array([sc["loss"], sc["mel_l2"], sc["kl"], sc["length_l2"]], np.float64)
        for k in sorted(g):
            out["ns2/" + tag + "/gdig/" + k] = digest(g[k])
            if tag == "kw1" and g[k].size <= 4096:
                out["ns2/kw1/grad/" + k] = g[k].astype(np.float32)
        if tag == "kw1":
            out["ns2/predictions"] = preds.astype(np.float32)
            for k, v in stats.items():
                out["ns2/moving/" + k] = v.astype(np.float64)
    outs, l2, kl, ll, _ = reference_call_training(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], eps_post, rf, 11)
    out.update({"ns2/call_l2": l2.astype(np.float64), "ns2/call_kl": kl.astype(np.float64), "ns2/call_length": ll.astype(np.float64)})
    return out


def build_ref_inverse():
    """tests/golden/refshim_inverse.npz -- Prior.Transformer.inverse = True (prior.py:81-99; BaseFlow.call / fwd_pass / bwd_pass swap
    _forward and _backward, flow.py:36-113), which neither LJHPS nor DataBakerHPS uses (hparams.py:344,462): the reference's own Python
    on the shims, tiny configuration: VAENAR.inference, VAENAR.call (evaluation mode), prior.sample / call / log_probability / init."""
    from oracle.run_reference_on_shim import reference_call, reference_inference, reference_module_methods
    hps = tiny_hps()
    hps.Prior.Transformer.inverse = True
    w = init_weights(hps, seed=SEED, mode="synthetic")
    B, Tt, Tm, rf = 3, 11, 40, 2
    b = make_batch(B, Tt, Tm, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, seed=19,
                   text_step=3, mel_step=7, temperature=1.0)
    r = np.random.Generator(np.random.PCG64(43))
    C, Tz = hps.Common.latent_dim, (Tm + rf - 1) // rf
    mels = r.standard_normal((B, Tm, hps.Audio.num_mels)).astype(np.float32)
    zl = ((b["mel_lengths"].astype(np.int64) + rf - 1) // rf).astype(np.int32)
    eps_post = r.standard_normal((B, 1, Tz, C)).astype(np.float32)
    eps_prior = r.standard_normal((B, int(zl.max()), C)).astype(np.float32)
    eps_init = r.standard_normal((B, int(zl.max()), C)).astype(np.float32)
    out = dict(ids=b["ids"], text_lengths=b["text_lengths"], mel_lengths=b["mel_lengths"], z_lengths=zl, mels=mels, eps=b["eps"],
               eps_post=eps_post, eps_prior=eps_prior, eps_init=eps_init, reduction_factor=np.int64(rf),
               weights_sha256=np.frombuffer(weights_digest(w).encode(), np.uint8))
    mel, ali, _, _ = reference_inference(hps, w, b["ids"], b["mel_lengths"], b["text_lengths"], b["eps"])
    out["mel"] = np.asarray(mel, np.float32)
    for k, v in ali.items():
        out["ali_" + k] = np.asarray(v, np.float32)
    outs, l2, kl, ll, _ = reference_call(hps, w, b["ids"], mels, b["mel_lengths"], b["text_lengths"], eps_post)
    out.update(call_outs=np.asarray(outs, np.float32), call_l2=np.asarray(l2, np.float64), call_kl=np.asarray(kl, np.float64),
               call_length=np.asarray(ll, np.float64))
    mm = reference_module_methods(hps, w, b["ids"], b["text_lengths"], mels[:, ::rf], zl, eps_post, eps_prior, eps_init, 11)
    for k, v in mm.items():
        if k.startswith("prior_") or k.startswith("init/") or k == "text_embd":
            out["mod/" + k] = np.asarray(v, np.float64)
    return out


def digest(a):
    """[16 samples at fixed strided flat positions | sum | l2 norm | max abs] of an array, float64."""
    f = np.asarray(a, np.float64).reshape(-1)
    idx = np.linspace(0, f.size - 1, 16).astype(np.int64)
    return np.concatenate([f[idx], [f.sum(), np.sqrt((f * f).sum()), np.abs(f).max()]])


def main():
    d = os.path.join(ROOT, "tests", "golden")
    os.makedirs(d, exist_ok=True)
    np.savez_compressed(os.path.join(d, "train_tiny.npz"), **build_train())
    print("wrote train_tiny", os.path.getsize(os.path.join(d, "train_tiny.npz")) // 1024, "KiB")
    for name in CASES:
        np.savez_compressed(os.path.join(d, name + ".npz"), **build(name))
        print("wrote", name, os.path.getsize(os.path.join(d, name + ".npz")) // 1024, "KiB")
    if os.path.isdir("/root/reference"):
        for name in REF_CASES:
            np.savez_compressed(os.path.join(d, name + ".npz"), **build_ref(name))
            print("wrote", name, os.path.getsize(os.path.join(d, name + ".npz")) // 1024, "KiB (reference's own Python over the tf shim)")
        np.savez_compressed(os.path.join(d, "refshim_nsample2.npz"), **build_ref_nsample(2))
        np.savez_compressed(os.path.join(d, "refshim_inverse.npz"), **build_ref_inverse())
        print("wrote refshim_inverse", os.path.getsize(os.path.join(d, "refshim_inverse.npz")) // 1024, "KiB")
        print("wrote refshim_nsample2", os.path.getsize(os.path.join(d, "refshim_nsample2.npz")) // 1024, "KiB (reference's own Python, num_samples = 2)")
        np.savez_compressed(os.path.join(d, "refshim_b2.npz"), **build_ref_b2())
        print("wrote refshim_b2", os.path.getsize(os.path.join(d, "refshim_b2.npz")) // 1024, "KiB (reference's own Python: module-level methods, num_samples = 2 training step)")
        for name in REF_TRAIN_CASES:
            np.savez_compressed(os.path.join(d, name + ".npz"), **build_ref_train(name))
            print("wrote", name, os.path.getsize(os.path.join(d, name + ".npz")) // 1024, "KiB (reference's own Python, training mode, over the torch tf shim)")


if __name__ == "__main__":
    main()
