#!/bin/bash
# round 5: raw stamp file of one S1 inference under the 4-wave kernel (for offline analysis): gpurun_out/cts_w4.bin
rm -f gpurun_out/cts_w4.bin
VNR_CHAIN_TS=gpurun_out/cts_w4.bin VNR_CHAIN_TS_STAGE=${1:-5} VNR_CHAIN_WAVES4=${W4:-1} VNR_CHAIN_PRIO=${PRIO:-1} python -c "
import sys; sys.path.insert(0,'.')
from vaenar_tts_amd.configs import LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights
w = init_weights(LJHPS, seed=1234, mode='synthetic', include_posterior=False)
m = VAENAR(LJHPS, device=0, weights=w)
b = make_batch(16, 128, 800, ragged=False, seed=1234, temperature=1.0)
m.inference(b['ids'], b['mel_lengths'], b['text_lengths'], reduction_factor=2, eps=b['eps'])
m.engine.synchronize()
"
ls -la gpurun_out/cts_w4.bin
