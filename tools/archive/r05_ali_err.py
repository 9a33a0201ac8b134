#!/usr/bin/env python3
"""round 5: decoder alignments of both chain kernels against the float64 oracle, absolute and RELATIVE (is the S^T / softmax path of the
4-wave kernel's attention phase less accurate?).  (Written while a rule kept alignment-writing launches on the 8-wave kernel;
VNR_CHAIN_W4_ALI lifted it.  Rule and switch are gone.)"""
import sys; sys.path.insert(0, '.')
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
from oracle.vaenar_numpy import Oracle
hps = LJHPS
w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
r64, rali = Oracle(hps, w, np.float64).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"].astype(np.float64))
m = VAENAR(hps, weights=w)
for w4 in (1, 0):
    m.engine.set_option("chain_waves4", w4)
    mel, ali = m.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
    print("chain_waves4=%d: mel err %.3e" % (w4, np.abs(mel.numpy() - r64).max()))
    for k in sorted(rali):
        a = rali[k]; g = ali[k].numpy().astype(np.float64)
        rel = np.abs(g - a) / a
        print("   %-20s abs err max %.2e   rel err max %.2e  median %.2e  p99 %.2e" % (k, np.abs(g - a).max(), rel.max(), np.median(rel), np.percentile(rel, 99)))
