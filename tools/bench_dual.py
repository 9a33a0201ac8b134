#!/usr/bin/env python3
"""Experiment: S1 batch as ONE stream of B=16 vs TWO concurrent streams of B=8 (two engines, two host threads)."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

w = init_weights(LJHPS, seed=1234, mode="synthetic", include_posterior=False)
steps = 20

def prep(model, B, seed):
    b = make_batch(B, 128, 800, seed=seed, temperature=1.0)
    e = model.engine
    return (e.to_device(b["ids"], np.int32), b["mel_lengths"], e.to_device(b["text_lengths"], np.int32), e.to_device(b["eps"], np.float32))

def run(model, args, n):
    for _ in range(n):
        model.inference(args[0], args[1], args[2], reduction_factor=2, eps=args[3])
    model.engine.synchronize()

for nstreams in (1, 2, 4):
    B = 16 // nstreams
    models = [VAENAR(LJHPS, weights=w) for _ in range(nstreams)]
    args = [prep(m, B, 1234 + i) for i, m in enumerate(models)]
    for m, a in zip(models, args): run(m, a, 3)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(m, a, steps)) for m, a in zip(models, args)]
    for t in th: t.start()
    for t in th: t.join()
    dt = (time.perf_counter() - t0) / steps
    print("streams=%d B/stream=%d  %.3f ms per 16 utterances  %.0f mel-frames/s" % (nstreams, B, dt * 1e3, 16 * 800 / dt))
    for m in models: m.engine.close()
