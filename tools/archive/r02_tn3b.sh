#!/bin/bash
# kernel-gradient GEMM: per-shape kernel time under measurement knobs (env list in $CASES, ';'-separated)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02r}; mkdir -p $O
IFS=';' read -ra CS <<< "$CASES"
i=0
for c in "${CS[@]}"; do
  i=$((i+1))
  export $c
  rocprofv3 --kernel-trace --stats -d $O/tr$i -o t -- python3 tools/bench_train.py 32 2 > $O/log$i.txt 2>&1
  python3 tools/rocpd_summary.py $(ls $O/tr$i/*.db $O/tr$i/*/*.db 2>/dev/null | head -1) > $O/stats$i.txt
  rm -rf $O/tr$i
  echo "== $c"; grep ms_per_step $O/log$i.txt | cut -c1-110; grep "gemm_tn3" $O/stats$i.txt | cut -c1-15,88-150
  for v in $c; do unset ${v%%=*}; done
done
