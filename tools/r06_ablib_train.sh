#!/bin/bash
# round 6: A/B of two builds of the library on ONE box, T1 training step (deterministic and atomics): the tree's libvaenar_hip.so against <other .so>
# usage: r06_ablib_train.sh <other .so> [reps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
other=$1; reps=${2:-2}
cp vaenar_tts_amd/libvaenar_hip.so /tmp/lib_tree.so
for rep in $(seq 1 $reps); do
  for which in tree other; do
    if [ $which = other ]; then cp $other vaenar_tts_amd/libvaenar_hip.so; else cp /tmp/lib_tree.so vaenar_tts_amd/libvaenar_hip.so; fi
    for det in 1 0; do
      VNR_TRAIN_OPTS="deterministic=$det" python tools/bench_train.py ${B:-32} 6 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-5s rep $rep det $det ms %.3f launches %.0f loss %.6f' % ('$which', d['ms_per_step'], d['launches_per_step'], d['loss']))"
    done
  done
done
cp /tmp/lib_tree.so vaenar_tts_amd/libvaenar_hip.so
