#!/usr/bin/env python3
"""round 4 experiment: one S1 batch (16 utterances) served as 1 x 16, 2 x 8 or 4 x 4 utterances on as many engine handles (streams),
all parts issued back to back from one host thread; prints ms per whole batch.  usage: r04_halves.py [opt=val ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
w = init_weights(LJHPS, seed=1234, mode='synthetic', include_posterior=False)
for parts in (1, 2, 4):
    B = 16 // parts
    lanes = []
    for i in range(parts):
        m = VAENAR(LJHPS, device=0, weights=w)
        for kv in sys.argv[1:]:
            k, v = kv.split("="); m.engine.set_option(k, int(v))
        b = make_batch(B, 128, 800, ragged=False, seed=1234 + i, temperature=1.0)
        lanes.append((m, m.engine.to_device(b["ids"], np.int32), m.engine.to_device(b["text_lengths"], np.int32), m.engine.to_device(b["eps"], np.float32), b))
    def step():
        for m, ids, tl, eps, b in lanes:
            m.inference(ids, b["mel_lengths"], tl, reduction_factor=2, eps=eps, return_alignments=True)
    for _ in range(3): step()
    for l in lanes: l[0].engine.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): step()
    for l in lanes: l[0].engine.synchronize()
    print("%d x %2d utterances on %d stream(s): %.3f ms per S1 batch" % (parts, B, parts, 1e3 * (time.perf_counter() - t0) / n))
    for l in lanes: l[0].engine.close()
