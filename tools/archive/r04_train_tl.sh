#!/bin/bash
# round 4: per-stream timeline of one T1 training step under the given environment; usage: r04_train_tl.sh <label> [ENV=..]...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
label=$1; shift
( for e in "$@"; do export $e; done
  rocprofv3 --kernel-trace -d $O/tt_$label -o t -- python3 tools/bench_train.py 32 3 > $O/tt_$label.log 2>&1 )
python3 tools/train_timeline.py $(ls $O/tt_$label/*.db $O/tt_$label/*/*.db 2>/dev/null | head -1) 4 > $O/train_tl_$label.txt
rm -rf $O/tt_$label
head -90 $O/train_tl_$label.txt | cut -c1-150
