#!/usr/bin/env python3
"""round 4: soak of the training step -- N steps at LJ widths with CHANGING batch shapes (the workspace arena, the activation-gradient
chunks, the deterministic mode's scratch and the event pool must reach a steady state), default and deterministic mode; prints losses,
step times and free device memory along the way.  usage: python tools/r04_soak.py [steps]"""
import sys, time
sys.path.insert(0, ".")
import ctypes as C
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hps = LJHPS
hip = C.CDLL("libamdhip64.so")


def free_mb():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2 ** 20


for det in (0, 1):
    model = VAENAR(hps, weights=init_weights(hps, seed=1234, mode="reference"))
    model.engine.set_option("deterministic", det)
    r = np.random.Generator(np.random.PCG64(7))
    shapes = [(32, 128, 800), (16, 96, 600), (32, 64, 400), (8, 128, 1000), (24, 100, 750)]
    t0 = time.time(); bad = 0; mem = []
    for i in range(steps):
        B, Tt, Tm = shapes[i % len(shapes)] if i % 7 else shapes[(i // 7) % len(shapes)]
        rf = (2, 3, 4, 5)[i % 4]
        b = make_batch(B, Tt, Tm, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, seed=i, text_step=1, mel_step=3)
        mels = r.standard_normal((B, Tm, 80)).astype(np.float32)
        model.prior.noise_seed, model.prior.noise_offset = 5, i << 26
        out = model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1e-5, rf, dropout_seed=i, learning_rate=1.25e-4)
        if not all(np.isfinite(x) for x in out):
            bad += 1
        if i % 25 == 0 or i == steps - 1:
            mem.append(free_mb())
            print("det %d step %3d  B %2d Tt %3d Tm %4d rf %d  loss %.4f l2 %.4f kl %.1f len %.4f  free %.0f MiB  %.1f ms/step avg" %
                  (det, i, B, Tt, Tm, rf, out[0], out[1], out[2], out[3], mem[-1], 1e3 * (time.time() - t0) / (i + 1)), flush=True)
    print("det %d: %d steps, non-finite %d, free memory first/last checkpoint %.0f / %.0f MiB (after step 25: %.0f)" % (det, steps, bad, mem[0], mem[-1], mem[1] if len(mem) > 1 else -1))
    assert bad == 0
    assert len(mem) < 3 or mem[-1] > mem[2] - 64, "device memory keeps shrinking: a leak"
    model.engine.close()
print("soak ok")
