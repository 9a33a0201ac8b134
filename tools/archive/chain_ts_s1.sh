rm -f /tmp/cts.bin
VNR_CHAIN_TS=/tmp/cts.bin VNR_CHAIN_ROWS64=1 python -c "
import sys; sys.path.insert(0,'.')
import numpy as np
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
w = init_weights(LJHPS, seed=1234, mode='synthetic', include_posterior=False)
m = VAENAR(LJHPS, device=0, weights=w)
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
m.inference(b['ids'], b['mel_lengths'], b['text_lengths'], reduction_factor=2, eps=b['eps'])
m.inference(b['ids'], b['mel_lengths'], b['text_lengths'], reduction_factor=2, eps=b['eps'])
m.engine.synchronize()
"
python tools/chain_timeline.py /tmp/cts.bin
