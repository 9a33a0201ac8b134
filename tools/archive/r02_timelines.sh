#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02c}; mkdir -p $O
for c in 8 4; do
  rm -f /tmp/a3.bin; VNR_ATTN3_CHUNKS=$c VNR_ATTN3_TS=/tmp/a3.bin python3 tools/s1_once.py > /dev/null 2>$O/a3_$c.err
  python3 tools/attn3_timeline.py /tmp/a3.bin > $O/attn3_timeline_chunks$c.txt 2>&1; cat $O/attn3_timeline_chunks$c.txt
done
for st in 1 8 0; do
  rm -f /tmp/cts.bin; VNR_CHAIN_TS=/tmp/cts.bin VNR_CHAIN_TS_STAGE=$st python3 tools/s1_once.py > /dev/null 2>$O/cts_$st.err
  python3 tools/chain_timeline.py /tmp/cts.bin > $O/chain_timeline_stage$st.txt 2>&1
  grep -A40 "stages=12" $O/chain_timeline_stage$st.txt | head -48
done
