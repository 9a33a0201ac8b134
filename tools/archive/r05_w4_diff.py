#!/usr/bin/env python3
"""round 5: where do the 4-wave and the 8-wave chain kernels differ?  One S1-shaped batch (ragged or full), both kernels, stage outputs compared."""
import sys; sys.path.insert(0, '.')
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
ragged = len(sys.argv) > 1 and sys.argv[1] == "ragged"
w = init_weights(LJHPS, seed=1234, mode='synthetic', include_posterior=False)
m = VAENAR(LJHPS, device=0, weights=w)
b = make_batch(16, 128, 800, ragged=ragged, seed=1234, temperature=1.0)
out = {}
z_ref = None
for w4 in (0, 1):
    m.engine.set_option("chain_waves4", w4)
    text = m.text_encoder(b["ids"], b["text_lengths"])
    zl = (np.asarray(b["mel_lengths"]) + 1) // 2
    z, _ = m.prior.sample(zl, text, b["text_lengths"], eps=b["eps"])
    z_ref = z.numpy() if z_ref is None else z_ref             # the decoder sees the SAME z under both kernels
    ini, outs, ali = m.decoder(z_ref, text, zl, b["text_lengths"], reduction_factor=2)
    out[w4] = dict(z=z.numpy(), ini=ini.numpy(), outs=outs.numpy(), **{k: v.numpy() for k, v in ali.items()})
for k in out[0]:
    a, c = out[0][k], out[1][k]
    d = np.abs(a - c)
    idx = np.unravel_index(np.argmax(d), d.shape)
    print("%-22s shape %-20s max|8w| %.3e  max diff %.3e at %s   mean diff %.3e" % (k, a.shape, np.abs(a).max(), d.max(), idx, d.mean()))
# per flow step: run the prior with fewer steps is not exposed; show the row profile of the z difference instead
d = np.abs(out[0]["z"] - out[1]["z"]).max(axis=2)
print("z diff by utterance (max over t):", np.array2string(d.max(axis=1), precision=2))
t = np.argmax(d, axis=1)
print("   at frame:", t)
