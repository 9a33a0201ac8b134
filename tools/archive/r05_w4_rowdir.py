#!/usr/bin/env python3
"""round 5: direction of the 4-wave kernel's row errors: is a bad row's error vector parallel to the row itself (a per-row scalar: LayerNorm
statistics, softmax sums) or not?"""
import sys; sys.path.insert(0, '.')
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
w = init_weights(LJHPS, seed=1234, mode='synthetic', include_posterior=False)
m = VAENAR(LJHPS, device=0, weights=w)
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
m.engine.set_option("chain_waves4", 0)
text = m.text_encoder(b["ids"], b["text_lengths"])
zl = (np.asarray(b["mel_lengths"]) + 1) // 2
z, _ = m.prior.sample(zl, text, b["text_lengths"], eps=b["eps"])
z = z.numpy()
out = {}
for w4 in (0, 1):
    m.engine.set_option("chain_waves4", w4)
    ini, outs, ali = m.decoder(z, text, zl, b["text_lengths"], reduction_factor=2)
    out[w4] = outs.numpy().reshape(16, 400, 160)
d = out[1] - out[0]
e = np.abs(d).max(axis=2)
bad = np.argwhere(e > 2e-5)
print("bad rows %d; worst %.2e" % (len(bad), e.max()))
mean_row = out[0].mean(axis=(0, 1))
for (u, t) in bad[np.argsort(-e[tuple(bad.T)])][:10]:
    dv = d[u, t]; xv = out[0][u, t] - mean_row
    cos = float(dv @ xv / (np.linalg.norm(dv) * np.linalg.norm(xv) + 1e-30))
    nb = [float(np.abs(d[u, tt]).max()) for tt in range(max(0, t - 2), min(400, t + 3))]
    print("utterance %2d row %3d (row mod 32 = %2d): |err| max %.2e  cos(err, row) %+.3f  |err|/|row| %.2e   neighbours' max err: %s" % (
        u, t, t % 32, np.abs(dv).max(), cos, np.linalg.norm(dv) / np.linalg.norm(xv), " ".join("%.1e" % v for v in nb)))
