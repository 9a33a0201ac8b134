#!/usr/bin/env python3
"""Time vnr_train_step on the T1 configuration (B=32, T_text=128, T_mel=800, rf in {2,5}),
synthetic data, random-init weights.  Prints one JSON line per reduction factor."""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
hps = LJHPS
model = VAENAR(hps, weights=init_weights(hps, seed=1234, mode="synthetic"))
import os
for kv in os.environ.get("VNR_TRAIN_OPTS", "").split():          # engine options for A/B runs, e.g. VNR_TRAIN_OPTS="gemm_wide_tiles=1"
    k, v = kv.split("=")
    model.engine.set_option(k, int(v))
b = make_batch(B, 128, 800, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=False)
r = np.random.Generator(np.random.PCG64(5))
mels = r.standard_normal((B, 800, 80)).astype(np.float32)
for rf in (2, 5):
    eps = r.standard_normal((B, (800 + rf - 1) // rf, 128)).astype(np.float32)
    ids = model.engine.asarray(b["ids"], np.int32); mel_d = model.engine.asarray(mels, np.float32); eps_d = model.engine.asarray(eps, np.float32)
    out = None
    for i in range(2):
        out = model.train_step(ids, mel_d, b["text_lengths"], b["mel_lengths"], 1e-5, rf, eps=eps_d, dropout_seed=i)
    model.engine.synchronize()
    n0 = model.engine.launch_count()
    t0 = time.perf_counter()
    for i in range(steps):
        out = model.train_step(ids, mel_d, b["text_lengths"], b["mel_lengths"], 1e-5, rf, eps=eps_d, dropout_seed=2 + i)
    model.engine.synchronize()          # (a step returns when its results are on the host; the kernel copies for the next step follow it)
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"workload": "T1 train_step B=%d T_text=128 T_mel=800 rf=%d" % (B, rf), "ms_per_step": dt * 1e3,
                      "mel_frames_per_s": B * 800 / dt, "launches_per_step": (model.engine.launch_count() - n0) / steps,
                      "loss": out[0], "mel_l2": out[1], "kl": out[2], "length_l2": out[3]}))
model.engine.close()
