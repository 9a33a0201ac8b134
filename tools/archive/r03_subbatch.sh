#!/bin/bash
# round 3 experiment: one S1 batch (16 utterances) served as sub-batches in flight on separate streams.
# ms_per_step x (16 / B) = time per 16 utterances
mkdir -p gpurun_out/r03a
run() { echo "== $*"; env "$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train --in-flight 0 "${@:2}" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'chain', d.get('roofline',{}).get('frac'))
"; }
run VNR_BENCH_B=16
run VNR_BENCH_B=8 --streams 2 --opt chain_rows64=0 --opt gemm_wide_tiles=0
run VNR_BENCH_B=8 --streams 2
run VNR_BENCH_B=4 --streams 4 --opt chain_rows64=0 --opt gemm_wide_tiles=0
run VNR_BENCH_B=4 --streams 4
run VNR_BENCH_B=16 --streams 2 --opt chain_rows64=0 --opt gemm_wide_tiles=0
