#!/bin/bash
# kernel-gradient GEMM alone: per-shape kernel time under measurement knobs (env list in $CASES, ';'-separated)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02s}; mkdir -p $O
IFS=';' read -ra CS <<< "$CASES"
i=0
for c in "${CS[@]}"; do
  i=$((i+1))
  for v in $c; do export $v; done
  rocprofv3 --kernel-trace --stats -d $O/tr$i -o t -- python3 tools/tn_bench.py 5 > $O/log$i.txt 2>&1
  python3 tools/rocpd_summary.py $(ls $O/tr$i/*.db $O/tr$i/*/*.db 2>/dev/null | head -1) > $O/stats$i.txt
  echo "== $c"; python3 tools/tn_per_shape.py $O/tr$i > $O/shape$i.txt; cat $O/shape$i.txt
  rm -rf $O/tr$i
  for v in $c; do unset ${v%%=*}; done
done
