#!/bin/bash
# training step: GPU train tests + timing (+ kernel stats when PROFILE=1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02t}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_train.py tests/test_refshim_train.py -m gpu -x -q > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log; tail -3 $O/pytest_train.log
python3 tools/bench_train.py 32 4 > $O/train_bench.txt 2>$O/train_bench.err; cut -c1-200 $O/train_bench.txt
if [ "$PROFILE" = "1" ]; then
  rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python3 tools/bench_train.py 32 3 > $O/train_prof.log 2>&1
  python3 tools/rocpd_summary.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats.txt
  rm -rf $O/tr
  head -24 $O/train_kernel_stats.txt | cut -c1-150
fi
