#!/usr/bin/env python3
"""The decoder cross-attention as a PHASE of the chain launch (round 5): one S1 inference with the chain kernel's
s_memtime stamps on (VNR_CHAIN_TS), the decoder blocks' launches picked by the header flag "attention + alignments", and the
attention + alignment phase (stamps 60 / 61 of csrc/gemm3.hip) reported as a share of the workgroup's lifetime.  Prints ONE JSON line:
{"launches": n, "phase_kcyc": median, "lifetime_kcyc": median, "share": phase / lifetime}.  Run as a child process by bench.py (the
stamp mode synchronises after every chain launch: never inside a timed region)."""
import json, os, struct, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

path = os.path.join(tempfile.mkdtemp(), "cts.bin")
os.environ["VNR_CHAIN_TS"] = path
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

w = init_weights(LJHPS, seed=1234, mode="synthetic", include_posterior=False)
m = VAENAR(LJHPS, device=0, weights=w)
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
for _ in range(2):                                       # (the first call also runs the range survey on the un-fused exact path)
    m.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
m.engine.synchronize()
open(path, "ab").close()
n0 = os.path.getsize(path)
m.inference(b["ids"], b["mel_lengths"], b["text_lengths"], reduction_factor=2, eps=b["eps"])
m.engine.synchronize()
data = open(path, "rb").read()[n0:]
off, ph, life = 0, [], []
while off < len(data):
    M, Dw, ns, nw = struct.unpack_from("4i", data, off); off += 16
    ts = np.frombuffer(data, dtype=np.uint64, count=nw * 128, offset=off).reshape(nw, 128).astype(np.int64); off += nw * 1024
    if (Dw >> 16) & 1 and (Dw >> 20):
        ph.append(float(np.median(ts[:, 61] - ts[:, 60]))); life.append(float(np.median(ts[:, 1 + 2 * ns] - ts[:, 0])))
out = {"launches": len(ph)}
if ph:
    out.update(phase_kcyc=float(np.mean(ph)) / 1e3, lifetime_kcyc=float(np.mean(life)) / 1e3, share=float(np.mean(ph) / np.mean(life)))
print(json.dumps(out))
