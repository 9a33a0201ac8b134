#!/bin/bash
# round 5: per-stage / per-wave s_memtime timelines of the chain launches of one S1 inference, 8-wave kernel against the 4-wave one.
# usage: r05_chain_ts.sh [stage with per-wave stamps]
st=${1:-5}
for w4 in 1 0; do
rm -f /tmp/cts.bin
VNR_CHAIN_TS=/tmp/cts.bin VNR_CHAIN_TS_STAGE=$st VNR_CHAIN_WAVES4=$w4 VNR_CHAIN_PRIO=${PRIO:-1} python -c "
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
echo "=== chain_waves4 = $w4, per-wave stamps of stage $st"
python tools/chain_timeline.py /tmp/cts.bin
done
