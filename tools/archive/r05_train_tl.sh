#!/bin/bash
# round 5: the training step's per-stream timeline with the stream -> hardware-queue mapping the un-profiled run has (GPU_MAX_HW_QUEUES=8
# exported in front of rocprofv3: the profiler's library initialises HIP before Python, where the engine's own setdefault comes too late)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05tl; mkdir -p $O
for q in 8 4; do
  export GPU_MAX_HW_QUEUES=$q
  VNR_TRAIN_OPTS="deterministic=${DET:-0}" rocprofv3 --kernel-trace --stats -d $O/tr$q -o t -- python3 tools/bench_train.py 32 3 > $O/train$q.log 2>&1
  python3 tools/train_timeline.py $(ls $O/tr$q/*.db $O/tr$q/*/*.db 2>/dev/null | head -1) 4 > $O/train_timeline_q$q.txt
  grep rf=2 $O/train$q.log | cut -c1-110
  sed -n 2,7p $O/train_timeline_q$q.txt | cut -c1-160
  grep -A6 "ten longest" $O/train_timeline_q$q.txt | cut -c1-120
  rm -rf $O/tr$q
done
