#!/usr/bin/env python3
"""Generate tests/golden/audio_lj.npz (test infrastructure): inputs and expected outputs of the vocoder step after the path
(reference audio/audio.py:81-102,232-246) from the float64 oracle oracle/audio_numpy.py.

    python oracle/make_audio_golden.py

The reference cannot produce these vectors here (librosa 0.8.0 is absent): PARITY UNPINNED, see oracle/audio_numpy.py.  The
fixture stores a small ragged case: mels [2, 24, 80] (second utterance 17 frames), the phase draw `2 pi rand` [2, 24, 1025] as
float32, and for each utterance the magnitudes' checksum, the waveform after 0 / 2 / 5 Griffin-Lim iterations and the de-emphasised
waveform of the 5-iteration result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.audio_numpy import AudioOracle  # noqa: E402
from vaenar_tts_amd.configs import LJHPS  # noqa: E402


def main():
    r = np.random.Generator(np.random.PCG64(20260930))
    lens = np.array([24, 17], np.int32)
    mels = r.uniform(0.05, 0.95, (2, 24, 80)).astype(np.float32)
    ang = (2 * np.pi * r.random((2, 24, 1025))).astype(np.float32)
    o = AudioOracle(LJHPS.Audio)
    out = {"mels": mels, "lengths": lens, "init_angles": ang}
    for b, n in enumerate(lens):
        mel = mels[b, :n].T.astype(np.float64)
        S = o.linear_from_mel(mel)
        out["S_sum_%d" % b] = np.array([S.sum(), (S ** 2).sum()])
        a = ang[b, :n].T.astype(np.float64)
        for it in (0, 2, 5):
            out["wav%d_it%d" % (b, it)] = o.griffin_lim(S, a, it).astype(np.float32)
        out["wav%d_deemph" % b] = o.inv_preemphasize(o.griffin_lim(S, a, 5)).astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "audio_lj.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
