#!/usr/bin/env python3
"""round 4: one kernel-gradient GEMM of the T1 step (M = 12800, K x N = 256 x 256 and 256 x 1024) with VNR_GEMM_TN3_TS set: per-phase timeline
(tools/tn3_timeline.py reads the dump).  usage: VNR_GEMM_TN3_TS=/tmp/tn.ts python tools/r04_tn3_tl.py"""
import sys; sys.path.insert(0, ".")
import numpy as np
from vaenar_tts_amd import _lib
from vaenar_tts_amd.configs import tiny_hps
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.weights import init_weights
hps = tiny_hps()
eng = VAENAR(hps, weights=init_weights(hps, seed=1)).engine
r = np.random.Generator(np.random.PCG64(1))
for (M, K, N) in ((12800, 256, 256), (12800, 256, 1024)):
    x = eng.to_device(r.standard_normal((M, K)).astype(np.float32)); dy = eng.to_device(r.standard_normal((M, N)).astype(np.float32))
    dw = eng.empty((K, N))
    for _ in range(2):
        _lib.check(eng.lib.vnr_op_kernel_grad(eng.handle, x.ptr, K, dy.ptr, N, M, K, N, 400, 0, dw.ptr), eng.handle)
    ref = x.numpy().astype(np.float64).T @ dy.numpy().astype(np.float64)
    print(M, K, N, "max rel err", float(np.abs(dw.numpy() - ref).max() / np.abs(ref).max()))
eng.close()
