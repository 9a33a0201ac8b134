#!/usr/bin/env python3
"""First-call cost of the range machinery at S1 size ("document the first-call latency"): a fresh engine's first inference
(exact-fp32 pass with the survey + the repeat on the split path), the steady state, and a call whose sentinel trips (replay on the exact
mode).  Host -> host, milliseconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
w = init_weights(LJHPS, seed=1234, mode="synthetic", include_posterior=False)
m = VAENAR(LJHPS, device=0, weights=w)
e = m.engine
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
ids, tl, eps = e.to_device(b["ids"]), e.to_device(b["text_lengths"]), e.to_device(b["eps"])
big = e.to_device((b["eps"].astype(np.float64) * 3.0e4).astype(np.float32))
def call(x):
    e.synchronize(); t = time.perf_counter()
    mel, _ = m.inference(ids, b["mel_lengths"], tl, reduction_factor=2, eps=x)
    mel.numpy()
    return 1e3 * (time.perf_counter() - t)
first = call(eps)
steady = min(call(eps) for _ in range(5))
trip = call(big)
after = min(call(eps) for _ in range(3))
print("first call (survey) %.2f ms | steady %.2f ms | call that trips the sentinel (split run + replay on the exact mode) %.2f ms | "
      "calls after the trip (modules on the exact mode) %.2f ms | %s" % (first, steady, trip, after, e.range_info()))
