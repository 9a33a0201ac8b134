#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02h}; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 > $O/pytest_full.log 2>&1; echo "rc=$?" >> $O/pytest_full.log
tail -40 $O/pytest_full.log
PROFILE=0 bash tools/r02_train.sh $O
