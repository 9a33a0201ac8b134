#!/bin/bash
# Round 6 baseline on the box: gemm per-workgroup timeline in the production configuration, launch sequence, quick bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
O=${1:-gpurun_out/r6base}; mkdir -p $O
P="--in-flight 0 --no-cpu-baseline --no-train --no-exact-pass --no-attn-phase"
python3 bench.py --steps 20 --warmup 5 $P > $O/bench.out 2> $O/bench.err; grep "^{" $O/bench.out | tail -1 > $O/bench.json
rm -f /tmp/g.ts; VNR_GEMM_TS=/tmp/g.ts python3 tools/s1_once.py > /dev/null 2>&1; python3 tools/gemm_timeline.py /tmp/g.ts > $O/gemm_timeline.txt 2>&1
rocprofv3 --kernel-trace -d $O/sq -o s -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 $P > $O/sq.log 2>&1; python3 tools/launch_sequence.py $(ls $O/sq/*.db $O/sq/*/*.db 2>/dev/null | head -1) > $O/launch_sequence.txt; rm -rf $O/sq
tail -c 400 $O/bench.json; echo; head -30 $O/gemm_timeline.txt
