#!/usr/bin/env python3
"""The measured floor of the decoder cross-attention kernel (VERDICT round 5 "next round" #5; north_star: ">= 40 % HBM on the decoder
cross-attention", reference /root/reference/modules/attention.py:224-246, decoder.py:192).  One S1 inference pass with the stand-alone
kernel (fuse_xattn = 0) under dispatch events, with attn3_kernel<true> built as a SKELETON (csrc/attention3.hip, VNR_ATTN3_SKEL -- read once
by the library, hence one child process per mode; bench.py runs this file three times):
  0  the kernel as it ships;
  1  traffic only: the 17.3 MB of operand images read, 13.1 MB of alignments + 6.55 MB of context written -- same grid, same loads, same
     store addresses, no MFMA, no softmax, no LDS transposes;
  2  all arithmetic and LDS traffic, no global store.
Prints ONE JSON line {"skel": m, "avg_launch_us": t, "launches": n}.  Results of modes 1 / 2 are meaningless numbers: the range machinery
is switched off for the pass."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

w = init_weights(LJHPS, seed=1234, mode="synthetic", include_posterior=False)
m = VAENAR(LJHPS, device=0, weights=w)
e = m.engine
for k, v in (("range_guard", 0), ("range_sentinel", 0), ("fuse_xattn", 0)):
    e.set_option(k, v)
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
ids, ml, tl, eps = e.to_device(b["ids"]), b["mel_lengths"], e.to_device(b["text_lengths"]), e.to_device(b["eps"])
for _ in range(3):
    m.inference(ids, ml, tl, reduction_factor=2, eps=eps)
e.synchronize()
e.profile(True); e.profile_reset()
for _ in range(5):
    m.inference(ids, ml, tl, reduction_factor=2, eps=eps)
e.synchronize()
p = e.profile_get("attn_cross_ali")
e.profile(False)
print(json.dumps({"skel": int(os.environ.get("VNR_ATTN3_SKEL", "0")), "launches": p["launches"],
                  "avg_launch_us": 1e3 * p["ms"] / max(1, p["launches"]), "bytes_per_launch": p["bytes"] / max(1, p["launches"])}))
