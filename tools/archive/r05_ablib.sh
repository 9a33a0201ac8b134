#!/bin/bash
# round 5: A/B of two builds of the library on ONE box: the tree's libvaenar_hip.so against gpurun_in/<name>.so; extra args go to bench.py
# usage: r04_ablib.sh <other .so> "<env assignments>" [bench args]
other=$1; envs=$2; shift 2
cp vaenar_tts_amd/libvaenar_hip.so /tmp/lib_tree.so
for rep in 1 2; do
  for which in tree other; do
    if [ $which = other ]; then cp $other vaenar_tts_amd/libvaenar_hip.so; else cp /tmp/lib_tree.so vaenar_tts_amd/libvaenar_hip.so; fi
    env $envs python bench.py --no-cpu-baseline --no-train --no-exact-pass --steps 20 --no-attn-phase --in-flight 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which rep $rep', 'ms %.3f' % d['ms_per_step'], 'in-flight-3 ms %.3f' % d.get('batches_in_flight_3', {}).get('ms_per_step', -1))"
  done
done
cp /tmp/lib_tree.so vaenar_tts_amd/libvaenar_hip.so
