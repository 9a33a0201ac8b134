#!/bin/bash
# kernel-gradient GEMM third generation: op test, train tests, T1 timing with and without it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02q}; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k kernel_gradient > $O/pytest_op.log 2>&1; tail -5 $O/pytest_op.log
timeout 900 python3 -m pytest tests/test_gpu_train.py tests/test_refshim_train.py -m gpu -x -q > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log; tail -3 $O/pytest_train.log
python3 tools/bench_train.py 32 4 > $O/train_tn3.txt 2>$O/train_tn3.err; cut -c1-120 $O/train_tn3.txt
VNR_GEMM_TN_V2=1 python3 tools/bench_train.py 32 4 > $O/train_tn2.txt 2>$O/train_tn2.err; cut -c1-120 $O/train_tn2.txt
for w in $WGS; do VNR_GEMM_TN3_WGS=$w python3 tools/bench_train.py 32 4 > $O/train_tn3_$w.txt 2>/dev/null; echo "wgs $w"; cut -c1-120 $O/train_tn3_$w.txt; done
if [ "$PROFILE" = "1" ]; then
  rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python3 tools/bench_train.py 32 3 > $O/train_prof.log 2>&1
  python3 tools/rocpd_summary.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats.txt
  rm -rf $O/tr
  head -12 $O/train_kernel_stats.txt | cut -c1-150; grep "gemm_tn" $O/train_kernel_stats.txt | cut -c1-150
fi
