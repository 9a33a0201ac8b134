#!/bin/bash
# same-box A/B of two builds of the library (gpurun_in/libA.so, gpurun_in/libB.so) on the T1 training step, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp vaenar_tts_amd/libvaenar_hip.so /tmp/lib_orig.so
for rep in 1 2 3; do for L in A B; do
  cp gpurun_in/lib$L.so vaenar_tts_amd/libvaenar_hip.so
  echo -n "lib$L "; python3 tools/bench_train.py 32 4 2>/dev/null | head -1 | cut -c50-110
done; done
cp /tmp/lib_orig.so vaenar_tts_amd/libvaenar_hip.so
