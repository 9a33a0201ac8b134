#!/bin/bash
# same-box A/B of two builds of the library (gpurun_in/libA.so, gpurun_in/libB.so): bench breakdown, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 3 --in-flight 0 --no-cpu-baseline --no-train"
cp vaenar_tts_amd/libvaenar_hip.so /tmp/lib_orig.so
for rep in 1 2 3; do for L in A B; do
  cp gpurun_in/lib$L.so vaenar_tts_amd/libvaenar_hip.so
  $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['end_to_end']['kernel_ms_per_step']
print('lib$L', 'ms/step %.3f' % d['ms_per_step'], 'chain %.3f gemm %.3f self %.3f' % (k['chain'], k['gemm'], k['attn_self']))"
done; done
cp /tmp/lib_orig.so vaenar_tts_amd/libvaenar_hip.so
