#!/bin/bash
# round 6: A/B of two builds of the library on ONE box -- the tree's libvaenar_hip.so against <other .so> (gpurun_in/...): S1 step and
# the kernel-class times of the profiled pass, alternating.   usage: r06_ablib.sh <other .so> [reps] [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
other=$1; reps=${2:-2}; shift 2
cp vaenar_tts_amd/libvaenar_hip.so /tmp/lib_tree.so
for rep in $(seq 1 $reps); do
  for which in tree other; do
    if [ $which = other ]; then cp $other vaenar_tts_amd/libvaenar_hip.so; else cp /tmp/lib_tree.so vaenar_tts_amd/libvaenar_hip.so; fi
    python bench.py --no-cpu-baseline --no-train --no-exact-pass --steps 20 --warmup 5 --no-attn-phase --in-flight 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['end_to_end']['kernel_ms_per_step']
print('%-5s rep $rep ms %.4f  chain %.4f gemm %.4f self %.4f ln %.4f misc %.4f launches %d h2h %.4f' % ('$which', d['ms_per_step'], k['chain']+k.get('chain_ali',0), k['gemm'], k['attn_self'], k['layer_norm'], k['misc'], d['end_to_end']['kernel_launches_per_step'], d['latency_host_to_host_ms']['median']))"
  done
done
cp /tmp/lib_tree.so vaenar_tts_amd/libvaenar_hip.so
