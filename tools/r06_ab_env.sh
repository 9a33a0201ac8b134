#!/bin/bash
# round 6: A/B of environment settings on ONE box.  usage: r06_ab_env.sh "<label>|<env assignments>" ...   (alternating REPS times, default 3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
for rep in $(seq 1 ${REPS:-3}); do
for cfg in "$@"; do
  IFS='|' read -r label envs <<< "$cfg"
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-exact-pass --in-flight 0 --no-attn-phase 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['end_to_end']['kernel_ms_per_step']
print('%-14s rep $rep ms %.4f  chain %.4f gemm %.4f self %.4f ln %.4f misc %.4f h2h %.4f err %.2e' % ('$label', d['ms_per_step'], k['chain']+k.get('chain_ali',0), k['gemm'], k['attn_self'], k['layer_norm'], k['misc'], d['latency_host_to_host_ms']['median'], d.get('parity',{}).get('max_abs_mel_err',-1)))"
done; done
