#!/usr/bin/env python3
"""Max-abs mel error of the S1 batch against the fp32 NumPy oracle for a few engine configurations (one handle each, and three
handles stepping concurrently)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.vaenar_numpy import Oracle
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

w = init_weights(LJHPS, seed=1234, mode="synthetic", include_posterior=False)
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
ref, _ = Oracle(LJHPS, w, np.float32).inference(b["ids"], b["mel_lengths"], b["text_lengths"], 2, b["eps"])
for opts in ({}, {"chain_rows64": 1}, {"chain_rows64": 1, "gemm_wide_tiles": 1}, {"chain_rows64": 1, "late_dec_kv": 0}):
    m = VAENAR(LJHPS, device=0, weights=w)
    for k, v in opts.items():
        m.engine.set_option(k, v)
    mel, _ = m.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
    print(opts, "max-abs err %.3e" % np.abs(mel.numpy() - ref).max())
    m.engine.close()
ms = [VAENAR(LJHPS, device=0, weights=w) for _ in range(3)]
for m in ms:
    m.engine.set_option("chain_rows64", 1)
    m.engine.set_option("gemm_wide_tiles", 1)
outs = []
for i in range(9):
    outs.append(ms[i % 3].inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])[0])
for m in ms:
    m.engine.synchronize()
print("3 handles concurrently:", ["%.3e" % np.abs(o.numpy() - ref).max() for o in outs])
