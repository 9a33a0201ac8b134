#!/bin/bash
# round 4: the launch sequence of one S1 step (rocprofv3 kernel trace) under the given environment; usage: r04_seq.sh <label> [ENV=..]...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
label=$1; shift
( for e in "$@"; do export $e; done
  rocprofv3 --kernel-trace -d $O/sq_$label -o s -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 --in-flight 0 --no-cpu-baseline --no-train --no-exact-pass > $O/sq_$label.log 2>&1 )
python3 tools/launch_sequence.py $(ls $O/sq_$label/*.db $O/sq_$label/*/*.db 2>/dev/null | head -1) > $O/seq_$label.txt
rm -rf $O/sq_$label
head -3 $O/seq_$label.txt
