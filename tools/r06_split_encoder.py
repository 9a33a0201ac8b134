#!/usr/bin/env python3
"""Round 6 experiment: the text encoder (33 launches at M = 2048 rows, every one under-filling the GPU) for ONE S1 batch as two half-batches on
two engine handles against one call.  usage: python tools/r06_split_encoder.py [steps] [parts]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hps = LJHPS
w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
B, Tt, Tm = 16, 128, 800
bt = make_batch(B, Tt, Tm, ragged=False, seed=1235, temperature=1.0)

def lane(lo, hi):
    m = VAENAR(hps, weights=w)
    e = m.engine
    return {"m": m, "ids": e.to_device(bt["ids"][lo:hi], np.int32), "tl": e.to_device(bt["text_lengths"][lo:hi], np.int32)}

def run(ln):
    return ln["m"].text_encoder(ln["ids"], ln["tl"], pos_step=2.795, training=False)

def timed(lanes, n):
    for ln in lanes: ln["m"].engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for ln in lanes: run(ln)
    for ln in lanes: ln["m"].engine.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

whole = [lane(0, B)]
for _ in range(5): run(whole[0])
ref = run(whole[0]).numpy()
t_whole = [timed(whole, steps) for _ in range(3)]
step = B // parts
split = [lane(i * step, (i + 1) * step) for i in range(parts)]
for _ in range(5):
    for ln in split: run(ln)
got = np.concatenate([run(ln).numpy() for ln in split], 0)
t_split = [timed(split, steps) for _ in range(3)]
print("encoder, one call of 16: %s ms   %d calls of %d on %d handles: %s ms   max |diff| %.3e" % (
    " ".join("%.3f" % t for t in t_whole), parts, step, parts, " ".join("%.3f" % t for t in t_split), float(np.abs(got - ref).max())))
