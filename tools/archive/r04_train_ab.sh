#!/bin/bash
# round 4: A/B of environment settings for the T1 training step on ONE box.  usage: r04_train_ab.sh "<label>|<env assignments>" ...
# prints ms per step at rf 2 / 5 for every setting, alternating twice
for rep in 1 2; do
for cfg in "$@"; do
  IFS='|' read -r label envs <<< "$cfg"
  out=$(env $envs python tools/bench_train.py 32 6 2>&1 | grep '^{' | python -c "
import sys, json
r = [json.loads(l) for l in sys.stdin]
print(' '.join('rf%s %.2f ms (%d launches)' % (x['workload'].split('rf=')[1], x['ms_per_step'], x['launches_per_step']) for x in r))")
  echo "$label rep $rep: $out"
done; done
