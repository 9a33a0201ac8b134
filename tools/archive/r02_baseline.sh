#!/bin/bash
# round-2 baseline on the GPU box: GPU tests, the default bench line, kernel stats of the same (single-stream) schedule, chain timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02a}; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python3 bench.py --steps 10 --warmup 3 --in-flight 0 --no-cpu-baseline --no-train > $O/kt.log 2>&1
python3 tools/rocpd_summary.py $(ls $O/kt/*.db $O/kt/*/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt
rm -rf $O/kt
head -30 $O/kernel_stats.txt | cut -c1-150
rm -f /tmp/cts.bin
VNR_CHAIN_TS=/tmp/cts.bin python3 -c "
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
python3 tools/chain_timeline.py /tmp/cts.bin > $O/chain_rows32_timeline.txt 2>&1
tail -60 $O/chain_rows32_timeline.txt
