#!/usr/bin/env python3
"""Two S1 inference calls on a fresh engine (timeline collection: VNR_CHAIN_TS / VNR_ATTN3_TS / VNR_GEMM_TS are read by the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
w = init_weights(LJHPS, seed=1234, mode='synthetic', include_posterior=False)
m = VAENAR(LJHPS, device=0, weights=w)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    m.engine.set_option(k, int(v))
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
for _ in range(2):
    m.inference(b['ids'], b['mel_lengths'], b['text_lengths'], reduction_factor=2, eps=b['eps'])
m.engine.synchronize()
