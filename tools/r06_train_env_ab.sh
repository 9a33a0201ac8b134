#!/bin/bash
# round 6: A/B of environment switches on the T1 training step (deterministic mode), alternating on ONE box.
# usage: r06_train_env_ab.sh "<label>|<env assignments>" ...    (REPS alternations, default 3; B = 32)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
for rep in $(seq 1 ${REPS:-3}); do
  for cfg in "$@"; do
    IFS='|' read -r label envs <<< "$cfg"
    env $envs VNR_TRAIN_OPTS="deterministic=${DET:-1}" python tools/bench_train.py ${B:-32} 6 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-13s rep $rep ms %.3f launches %.0f loss %.6f' % ('$label', d['ms_per_step'], d['launches_per_step'], d['loss']))"
  done
done
