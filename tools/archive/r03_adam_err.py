#!/usr/bin/env python3
"""Round 3 diagnosis: the step-2 gradient check of tests/test_gpu_train.py::test_adam_update_matches_keras_formula, repeated.
Prints the worst relative gradient errors (against the float64 autograd oracle) of every repetition."""
import os, sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
from test_gpu_train import _case
from oracle.vaenar_torch import TorchOracle
from vaenar_tts_amd.models import VAENAR

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
hps, w, b, mels, eps = _case("tiny")
ref = None
for rep in range(reps):
    model = VAENAR(hps, weights=w)
    import os
    for kv in os.environ.get("VNR_TRAIN_OPTS", "").split():
        k_, v_ = kv.split("="); model.engine.set_option(k_, int(v_))
    try:
        for step in (1, 2):
            before = model.get_weights()
            out = model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], 1.0, 2, eps=eps, dropout_seed=step,
                             learning_rate=float(os.environ.get('LR', '1e-3')), apply_update=True)
            g = {k: x.astype(np.float64) for k, x in model.gradients().items()}
    finally:
        model.engine.close()
    if ref is None:
        ref, _ = TorchOracle(hps, before).gradients(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, kl_weight=1.0, dropout_seed=2)
    errs = sorted(((np.abs(g[k] - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-30), k) for k in ref), reverse=True)
    kk = "prior/glow/1/2/net/attentions/1/ffn/dense1/kernel"
    print("KEY %.2e" % (np.abs(g[kk] - ref[kk]).max() / np.abs(ref[kk]).max()), flush=True)
    kb = "prior/glow/1/2/net/attentions/1/ffn/dense1/bias"
    eb = np.abs(g[kb] - ref[kb]) / np.abs(ref[kb]).max()
    print("BIAS elements off by > 1e-4 of the max: %d of %d (where %s, errors %s)" % ((eb > 1e-4).sum(), eb.size, np.nonzero(eb > 1e-4)[0][:5], eb[eb > 1e-4][:5]), flush=True)
    print("WORST %.1e" % errs[1][0], flush=True)
    print("rep %d: scalars %r " % (rep, tuple(float(x) for x in out[:4])) + "  ".join("%.2e %s" % (e, k[-60:]) for e, k in errs[:3]), flush=True)
    if errs[1][0] > 1e-4:
        for e, k in errs[1:16]: print("      %.2e %s" % (e, k))
