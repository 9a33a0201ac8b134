#!/usr/bin/env python3
"""round 5 debugging aid (needs the VNR_DBG_CTX patch of that day -- a float* dbg_ctx in ChainArgs filled by the kernel and dumped by
launch_panel_chain; removed again, see profiles/r05_experiments.txt r05i): the decoder on the SAME z under one chain kernel, with VNR_DBG_CTX dumping the attention context of its two
block launches.  usage: VNR_DBG_CTX=<file> r05_ctx_dump.py <chain_waves4> <outs.npy>"""
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
m.engine.set_option("chain_waves4", int(sys.argv[1]))
if len(sys.argv) > 3: m.engine.set_option("fuse_xattn", int(sys.argv[3]))
ini, outs, ali = m.decoder(z, text, zl, b["text_lengths"], reduction_factor=2)
np.save(sys.argv[2], outs.numpy())
