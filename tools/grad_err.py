import sys; sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_train import _case
from oracle.vaenar_torch import TorchOracle
from vaenar_tts_amd.models import VAENAR
name, kw = sys.argv[1], float(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
hps, w, b, mels, eps = _case(name)
model = VAENAR(hps, weights=w)
model.train_step(b["ids"], mels, b["text_lengths"], b["mel_lengths"], kw, 2, eps=eps, dropout_seed=seed, apply_update=False)
got = model.gradients(); model.engine.close()
o = TorchOracle(hps, w)
ref, sc = o.gradients(b["ids"], mels, b["mel_lengths"], b["text_lengths"], 2, eps, kl_weight=kw, length_weight=hps.Train.length_weight, dropout_seed=seed)
rows = sorted(((float(np.abs(got[k] - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-30)), k, float(np.abs(ref[k]).max())) for k in ref if np.abs(ref[k]).max() > 0), reverse=True)
for r in rows[:8]: print("%.2e  %-70s |ref|max %.3e" % r)
print("median rel err %.2e, tensors above 1e-3: %d, above 1e-4: %d; closest FFN units to the ReLU kink: %s" % (np.median([r[0] for r in rows]), sum(r[0] > 1e-3 for r in rows), sum(r[0] > 1e-4 for r in rows), sorted(o.last.get("relu_margin", {}).values())[:3]))
