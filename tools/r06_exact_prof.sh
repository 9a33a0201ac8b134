#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r6exact; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python3 tools/s1_once.py split_fp16=0 > $O/kt.log 2>&1
python3 tools/rocpd_summary.py $(ls $O/kt/*.db $O/kt/*/*.db 2>/dev/null | head -1) | head -24 | cut -c1-170
rm -rf $O/kt
