#!/usr/bin/env python3
"""Round 6 debugging aid: where does the exact-fp32 mode lose accuracy when the injected prior noise is large?  Module by module against
the float64 oracle (text encoder -> prior.sample -> decoder), split_fp16 = 0, scales 1 ... 3e5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.vaenar_numpy import Oracle
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
hps = LJHPS
w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
b = make_batch(4, 37, 150, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, temperature=1.0, text_step=5, mel_step=23)
o = Oracle(hps, w, np.float64)
m = VAENAR(hps, weights=w)
m.engine.set_option("split_fp16", int(os.environ.get("SPLIT", "0")))
pos_step = np.float32(hps.Common.mel_text_len_ratio) / np.float32(2)
red = ((b["mel_lengths"].astype(np.int64) + 1) // 2).astype(np.int32)
te = m.text_encoder(b["ids"], b["text_lengths"], pos_step=pos_step)
rte = o.text_encoder(b["ids"], b["text_lengths"], pos_step=pos_step)
print("text encoder err %.3e" % np.abs(te.numpy() - rte).max())
for sc in (1.0, 1e2, 1e4, 3e5):
    eps = (b["eps"].astype(np.float64) * sc).astype(np.float32)
    z, lp = m.prior.sample(red, te, b["text_lengths"], eps=eps)
    rz, rlp = o.prior_sample(red, rte, b["text_lengths"], eps.astype(np.float64))
    zz = z.numpy()
    print("scale %.0e: z finite %s  max|z| %.3g  rel err %.3e" % (sc, np.isfinite(zz).all(), np.abs(rz).max(), np.abs(zz - rz).max() / np.abs(rz).max()))
    _, mel, ali = m.decoder(inputs=z, text_embd=te, z_lengths=red, text_lengths=b["text_lengths"], training=False, reduction_factor=2)
    _, rmel, rali = o.decoder(rz, rte, red, b["text_lengths"], 2)
    mm = mel.numpy()
    print("            mel finite %s  err %.3e (max|mel| %.3g)  nan count %d" % (np.isfinite(mm).all(), np.nanmax(np.abs(mm - rmel)), np.abs(rmel).max(), np.isnan(mm).sum()))
    # decoder fed with the ORACLE's z: isolates the decoder
    _, mel2, _ = m.decoder(inputs=rz.astype(np.float32), text_embd=te, z_lengths=red, text_lengths=b["text_lengths"], training=False, reduction_factor=2)
    print("            decoder alone (oracle z): err %.3e finite %s" % (np.nanmax(np.abs(mel2.numpy() - rmel)), np.isfinite(mel2.numpy()).all()))
