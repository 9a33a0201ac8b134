#!/bin/bash
# round 3: kernel trace of the T1 training step -> per-stream timeline + kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r03t}; mkdir -p $O
python3 tools/bench_train.py 32 4 2>&1 | cut -c1-150 | tee $O/train_bench.txt
rocprofv3 --kernel-trace -d $O/tr -o t -- python3 tools/bench_train.py 32 3 > $O/train_prof.log 2>&1
DB=$(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1)
python3 tools/rocpd_summary.py $DB > $O/train_kernel_stats.txt
python3 tools/train_timeline.py $DB 4 > $O/train_timeline.txt      # (bench_train: 2 + 3 steps at rf = 2, then 2 + 3 at rf = 5)
python3 tools/train_timeline.py $DB -2 > $O/train_timeline_rf5.txt
rm -rf $O/tr
head -60 $O/train_timeline.txt
