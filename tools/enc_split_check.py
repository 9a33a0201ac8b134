import numpy as np, sys
sys.path.insert(0, '/root/repo')
from oracle.vaenar_numpy import Oracle
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
w = init_weights(LJHPS, seed=1234, include_posterior=False)
o = Oracle(LJHPS, w, np.float64)
m = VAENAR(LJHPS, weights=w)
for mode in (0, 1):
    m.engine.set_option("split_encoder", mode)
    worst = 0; flips = 0; n = 0; minmargin = 1
    for seed in range(6):
        b = make_batch(16, 64, 100, ragged=True, seed=seed, text_step=3)
        pos = np.float32(5.59) / np.float32(2)
        te = m.text_encoder(b["ids"], b["text_lengths"], pos_step=pos)
        pl = m.length_predictor(te, b["text_lengths"]).numpy()
        rte = o.text_encoder(b["ids"], b["text_lengths"], pos_step=pos)
        rl = o.length_predictor(rte, b["text_lengths"])
        worst = max(worst, np.abs(te.numpy() - rte).max())
        margin = np.minimum(rl - np.floor(rl), np.ceil(rl) - rl)
        flips += int((pl.astype(np.int32) != rl.astype(np.float32).astype(np.int32)).sum()); n += len(pl)
        minmargin = min(minmargin, margin.min())
        relerr = np.abs(pl - rl).max() / rl.max()
    print("split_encoder=%d  text_embd max err %.2e  length rel err %.1e  integer flips %d / %d  (min margin %.3g)" % (mode, worst, relerr, flips, n, minmargin))
