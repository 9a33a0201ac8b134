#!/bin/bash
# round 6: the deterministic training step with the ordered finish inside gemm_tn3_kernel against the round-4 second launch
# (VNR_DET_SEPARATE_FINISH=1), and the float-atomics mode, alternating on ONE box; B = 32 and B = 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
  for cfg in "det_inkernel|deterministic=1|" "det_separate|deterministic=1|VNR_DET_SEPARATE_FINISH=1" "atomics|deterministic=0|"; do
    IFS='|' read -r label opts envs <<< "$cfg"
    for B in 32 4; do
      env $envs VNR_TRAIN_OPTS="$opts" python tools/bench_train.py $B 6 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-13s rep $rep B=%-2d ms %.3f launches %.0f loss %.6f' % ('$label', $B, d['ms_per_step'], d['launches_per_step'], d['loss']))"
    done
  done
done
