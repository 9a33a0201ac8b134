#!/bin/bash
# round 3: per-kernel stats of the attention backward under A/B switches: "label|ENV=1 ENV2=1" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
for spec in "$@"; do
  v=${spec%%|*}; envs=${spec#*|}
  ( for e in $envs; do export $e; done
    rocprofv3 --kernel-trace -d $O/tr_$v -o t -- python3 tools/bench_train.py 32 2 > $O/prof_$v.log 2>&1 )
  python3 tools/rocpd_summary.py $(ls $O/tr_$v/*.db $O/tr_$v/*/*.db 2>/dev/null | head -1) > $O/stats_$v.txt
  rm -rf $O/tr_$v
  echo "== $v"; grep -E "attn_bwd" $O/stats_$v.txt | grep grid | cut -c1-170 | head -24
done
