#!/usr/bin/env python3
"""round 5: S1 mel error against the float64 oracle under the chain-kernel switches (which feature of the 4-wave kernel costs accuracy?)."""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
from oracle.vaenar_numpy import Oracle
hps = LJHPS
w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
r64, _ = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"].astype(np.float64))
m = VAENAR(hps, weights=w)
def run(tag, ali=True, **opts):
    for k, v in opts.items(): m.engine.set_option(k, v)
    mel, _ = m.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"], return_alignments=ali)
    print("%-46s err vs float64 oracle %.3e" % (tag, np.abs(mel.numpy() - r64).max()), flush=True)
    for k in opts: m.engine.set_option(k, {"chain_waves4": 1, "fuse_xattn": 1, "chain_segments": 1, "chain_prefetch": 1}[k])
run("4-wave kernel (default)")
sys.exit(0) if os.environ.get("ONLY_DEFAULT") else None
run("4-wave kernel, alignments not requested", ali=False)
run("8-wave kernel", chain_waves4=0)
run("4-wave, no fused attention", fuse_xattn=0)
run("4-wave, flat panels", chain_segments=0)
run("4-wave, no prefetch workgroups", chain_prefetch=0)
mel, _ = m.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
d = np.abs(mel.numpy() - r64)
print("error by utterance (max):", np.array2string(d.max(axis=(1, 2)), precision=1))
fr = d.max(axis=(0, 2))                       # by mel frame (2 frames per latent row)
top = np.argsort(fr)[-12:][::-1]
print("largest errors at mel frames:", top.tolist(), "->", np.array2string(fr[top], precision=1))
print("mean error %.2e, median %.2e, fraction of elements above 2e-5: %.4f" % (d.mean(), np.median(d), (d > 2e-5).mean()))
lat = fr.reshape(-1, 2).max(axis=1)           # by latent row
print("by latent row mod 32 (max):", np.array2string(np.array([lat[i::32].max() for i in range(32)]), precision=1))
bad = d > 2e-5
print("bad elements %d; mel frames with any: %d of %d; frames fully bad (>=40 of 80 bins): %d" % (bad.sum(), bad.any(axis=2).sum(), bad.shape[0] * bad.shape[1], (bad.sum(axis=2) >= 40).sum()))
print("bad per utterance:", bad.sum(axis=(1, 2)).tolist())
print("bad per mel bin:", bad.sum(axis=(0, 1)).tolist())
rows = np.argwhere(bad.any(axis=2))
print("first bad (utterance, frame):", rows[:40].tolist())
