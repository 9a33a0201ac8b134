import sys; sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_train import _case
from oracle.vaenar_torch import TorchOracle
from vaenar_tts_amd.models import VAENAR
name, kw = sys.argv[1], float(sys.argv[2])
hps, w, b, mels, eps = _case(name)
model = VAENAR(hps, weights=w)
model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], kw, 2, eps=eps, dropout_seed=11, apply_update=False)
got = model.gradients(); model.engine.close()
ref, sc = TorchOracle(hps, w).gradients(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, kl_weight=kw, length_weight=hps.Train.length_weight, dropout_seed=11)
rows = sorted(((float(np.abs(got[k] - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-30)), k, float(np.abs(ref[k]).max())) for k in ref if np.abs(ref[k]).max() > 0), reverse=True)
for r in rows[:8]: print("%.2e  %-70s |ref|max %.3e" % r)
print("median rel err %.2e" % np.median([r[0] for r in rows]))
