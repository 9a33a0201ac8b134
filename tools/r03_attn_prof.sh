#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
for v in new stored; do
  [ $v = stored ] && export VNR_ATTN_BWD_STORED=1
  rocprofv3 --kernel-trace -d $O/tr_$v -o t -- python3 tools/bench_train.py 32 2 > $O/prof_$v.log 2>&1
  python3 tools/rocpd_summary.py $(ls $O/tr_$v/*.db $O/tr_$v/*/*.db 2>/dev/null | head -1) > $O/stats_$v.txt
  rm -rf $O/tr_$v
  echo "== $v"; grep -E "attn" $O/stats_$v.txt | cut -c1-170 | head -24
done
