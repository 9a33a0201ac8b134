#!/usr/bin/env python3
"""Round 6 experiment: ONE S1 batch (16 utterances) as two half-batches of 8 on two engine handles (own stream, workspace and weight copy
each), issued back to back and joined at the end -- against the same 16 utterances as one call.  Prints ms per 16 utterances.
usage: python tools/r06_split_batch.py [steps] [parts]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hps = LJHPS
w = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
B, Tt, Tm, rf = 16, 128, 800, 2
bt = make_batch(B, Tt, Tm, ragged=False, seed=1235, temperature=1.0)

def lane(lo, hi, opts=()):
    m = VAENAR(hps, weights=w)
    for k, v in opts: m.engine.set_option(k, v)
    e = m.engine
    return {"m": m, "ids": e.to_device(bt["ids"][lo:hi], np.int32), "tl": e.to_device(bt["text_lengths"][lo:hi], np.int32),
            "eps": e.to_device(bt["eps"][lo:hi], np.float32), "ml": bt["mel_lengths"][lo:hi]}

def run(ln):
    return ln["m"].inference(ln["ids"], ln["ml"], ln["tl"], reduction_factor=rf, eps=ln["eps"], return_alignments=True)

def timed(lanes, n):
    for ln in lanes: ln["m"].engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for ln in lanes: run(ln)
    for ln in lanes: ln["m"].engine.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

whole = [lane(0, B)]
for _ in range(5): run(whole[0])
ref = run(whole[0])[0].numpy()
t_whole = [timed(whole, steps) for _ in range(3)]
step = B // parts
split = [lane(i * step, (i + 1) * step) for i in range(parts)]
for _ in range(5):
    for ln in split: run(ln)
got = np.concatenate([run(ln)[0].numpy() for ln in split], 0)
t_split = [timed(split, steps) for _ in range(3)]
t_whole2 = [timed(whole, steps) for _ in range(2)]
print("one call of 16: %s ms   %d calls of %d on %d handles: %s ms   (one call again: %s)   max |mel diff| %.3e" % (
    " ".join("%.3f" % t for t in t_whole), parts, step, parts, " ".join("%.3f" % t for t in t_split), " ".join("%.3f" % t for t in t_whole2),
    float(np.abs(got - ref).max())))
