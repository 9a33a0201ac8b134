#!/usr/bin/env python3
"""Kernel-gradient GEMM alone (vnr_op_kernel_grad) on the T1 shapes; run under rocprofv3 --kernel-trace --stats and read the
per-shape kernel durations (the op itself synchronises and allocates: wall time is not the figure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vaenar_tts_amd import _lib
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.weights import init_weights
m = VAENAR(LJHPS, device=0, weights=init_weights(LJHPS, seed=1, mode='synthetic', include_posterior=False))
eng = m.engine
r = np.random.default_rng(0)
shapes = [(12800, 256, 256), (12800, 512, 256), (12800, 256, 1024), (12800, 1024, 256), (25600, 512, 512), (4096, 256, 256)]
if os.environ.get('TN_SHAPES'):
    shapes = [tuple(int(v) for v in t.split('x')) for t in os.environ['TN_SHAPES'].split(',')]
for (M, K, N) in shapes:
    x = eng.asarray(r.standard_normal((M, K)).astype(np.float32), np.float32)
    dy = eng.asarray(r.standard_normal((M, N)).astype(np.float32), np.float32)
    dw = eng.empty((K, N))
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        _lib.check(eng.lib.vnr_op_kernel_grad(eng.handle, x.ptr, K, dy.ptr, N, M, K, N, M, 0, dw.ptr), eng.handle)
    print('shape', M, K, N, flush=True)
