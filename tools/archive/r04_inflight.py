#!/usr/bin/env python3
"""round 4 experiment: N S1 batches in flight on N engine handles (64-row chain panels, 64x128 GEMM tiles), issued round-robin from one
host thread -- the bench's batches_in_flight block without anything around it.  usage: r04_inflight.py [N=3] [opt=val ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
opts = [a for a in sys.argv[1:] if "=" in a] or ["chain_rows64=1", "gemm_wide_tiles=1"]
w = init_weights(LJHPS, seed=1234, mode='synthetic', include_posterior=False)
if "--pre" in sys.argv:          # like bench.py: a default-option engine runs first and stays alive
    m0 = VAENAR(LJHPS, device=0, weights=w)
    b0 = make_batch(16, 128, 800, ragged=False, seed=1, temperature=1.0)
    for _ in range(25):
        m0.inference(b0["ids"], b0["mel_lengths"], b0["text_lengths"], reduction_factor=2, eps=b0["eps"], return_alignments=True)
    m0.engine.synchronize()
lanes = []
for i in range(N):
    m = VAENAR(LJHPS, device=0, weights=w)
    for kv in opts:
        k, v = kv.split("="); m.engine.set_option(k, int(v))
    b = make_batch(16, 128, 800, ragged=False, seed=1234 + i, temperature=1.0)
    lanes.append((m, m.engine.to_device(b["ids"], np.int32), m.engine.to_device(b["text_lengths"], np.int32), m.engine.to_device(b["eps"], np.float32), b))
def step(i):
    m, ids, tl, eps, b = lanes[i % N]
    m.inference(ids, b["mel_lengths"], tl, reduction_factor=2, eps=eps, return_alignments=True)
for i in range(3 * N): step(i)
for l in lanes: l[0].engine.synchronize()
n = 30
t0 = time.perf_counter()
for i in range(n): step(i)
t_issue = time.perf_counter() - t0
for l in lanes: l[0].engine.synchronize()
print("%d batches in flight (%s): %.3f ms per S1 batch; host issue %.3f ms per batch" % (N, " ".join(opts), 1e3 * (time.perf_counter() - t0) / n, 1e3 * t_issue / n))
