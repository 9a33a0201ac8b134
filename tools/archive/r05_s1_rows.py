#!/usr/bin/env python3
"""round 5: what do the latent rows with a large mel error under the 4-wave chain kernel have in common?  (oracle attention statistics)"""
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
mel, ali = m.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
d = np.abs(mel.numpy() - r64).max(axis=2).reshape(16, 400, 2).max(axis=2)      # [utterance][latent row]
bad = d > 2e-5
print("bad latent rows: %d of %d" % (bad.sum(), bad.size))
for k in sorted(rali):
    a = rali[k]                                    # [B, H, Tq, Tk]
    pmax = a.max(axis=3).max(axis=1)               # most peaked head of the row
    pmin = a.max(axis=3).min(axis=1)
    ent = -(a * np.log2(np.maximum(a, 1e-300))).sum(axis=3).min(axis=1)
    g = np.abs(ali[k].numpy() - a).max(axis=3).max(axis=1)
    print("%-22s p_max of the most peaked head: bad rows median %.4f (min %.4f)  all rows median %.4f;  lowest entropy: bad %.3f all %.3f bits;  alignment err: bad rows %.2e all %.2e"
          % (k, np.median(pmax[bad]), pmax[bad].min(), np.median(pmax), np.median(ent[bad]), np.median(ent), g[bad].max(), g.max()))
