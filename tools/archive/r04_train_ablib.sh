#!/bin/bash
# round 4: A/B of two library builds for the T1 training step on ONE box: the tree's libvaenar_hip.so against <other .so>
# usage: r04_train_ablib.sh <other .so> ["<env assignments>"]
other=$1; envs=${2:-A=1}
cp vaenar_tts_amd/libvaenar_hip.so /tmp/lib_tree.so
for rep in 1 2; do
  for which in tree other; do
    if [ $which = other ]; then cp $other vaenar_tts_amd/libvaenar_hip.so; else cp /tmp/lib_tree.so vaenar_tts_amd/libvaenar_hip.so; fi
    out=$(env $envs python tools/bench_train.py 32 6 2>&1 | grep '^{' | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin]
print(' '.join('rf%s %.2f ms' % (x['workload'].split('rf=')[1], x['ms_per_step']) for x in r))")
    echo "$which rep $rep ($envs): $out"
  done
done
cp /tmp/lib_tree.so vaenar_tts_amd/libvaenar_hip.so
