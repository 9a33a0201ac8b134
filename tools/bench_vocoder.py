#!/usr/bin/env python3
"""Griffin-Lim vocoder step on the S1 batch shape (16 utterances x 800 mel frames, 60 iterations, n_fft 2048 / hop 256 / win 1024):
GPU time per batch through the C ABI, next to the NumPy oracle (restated librosa) on one utterance with fewer iterations."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaenar_tts_amd import _lib  # noqa: E402
from vaenar_tts_amd.audio import Audio  # noqa: E402
from vaenar_tts_amd.configs import LJHPS, tiny_hps  # noqa: E402


def main():
    B, T = 16, 800
    eng = _lib.Engine(tiny_hps(), 0)
    au = Audio(LJHPS.Audio, engine=eng)
    r = np.random.Generator(np.random.PCG64(0))
    mels = eng.to_device(r.uniform(0, 1, (B, T, 80)).astype(np.float32))
    S = au.linear_from_mel_batch(mels)
    for iters in (60,):
        ts = []
        for _ in range(6):                               # the first calls pay workspace growth and clock ramp: report the best
            eng.synchronize()
            t0 = time.perf_counter()
            au._griffin_lim_batch(S, None, None, seed=1, n_iters=iters)
            eng.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = min(ts[2:])
        sec_audio = B * 256 * (T - 1) / 22050.0
        print("griffin_lim B=%d T=%d iters=%d: %.2f ms per batch = %.1f us per iteration, %.0f x real time (%.1f s of audio)"
              % (B, T, iters, 1e3 * dt, 1e6 * dt / (iters + 1), sec_audio / dt, sec_audio))
    t0 = time.perf_counter(); au.linear_from_mel_batch(mels); eng.synchronize()
    print("mel_to_linear: %.3f ms" % (1e3 * (time.perf_counter() - t0)))
    if "--cpu" in sys.argv:
        from oracle.audio_numpy import AudioOracle
        o = AudioOracle(LJHPS.Audio)
        Sh = S.numpy()[0].T.astype(np.float64)
        ang = 2 * np.pi * r.random(Sh.shape)
        t0 = time.perf_counter(); o.griffin_lim(Sh, ang, 5); d = time.perf_counter() - t0
        print("NumPy oracle (pocketfft, 1 thread), 1 utterance, 5 iterations: %.2f s => %.1f s per 60-iteration utterance, %.0f s per batch"
              % (d, d * 61 / 6, d * 61 / 6 * B))
    eng.close()


if __name__ == "__main__":
    main()
