#!/bin/bash
# A/B of the round-2 kernel changes: persistent attn3 (chunks per pair) and the straight-line chain k-loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02b}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_golden.py -m gpu -x -q > $O/pytest_core.log 2>&1; echo "rc=$?" >> $O/pytest_core.log; tail -3 $O/pytest_core.log
B="python3 bench.py --steps 20 --warmup 3 --in-flight 0 --no-cpu-baseline --no-train"
for c in default 0 4 6 13; do
  if [ $c = default ]; then unset VNR_ATTN3_CHUNKS; else export VNR_ATTN3_CHUNKS=$c; fi
  $B > $O/bench_chunks_$c.json 2> $O/bench_chunks_$c.err
  python3 - $O/bench_chunks_$c.json $c <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["end_to_end"]["kernel_ms_per_step"]
print("chunks", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "chain %.3f gemm %.3f self %.3f cross %.3f ali_us %.2f" % (k["chain"],k["gemm"],k["attn_self"],k["attn_cross"], d["roofline_cross_attention"]["avg_launch_us"]), "frac_x %.3f" % d["roofline_cross_attention"]["frac"])
PY
done
unset VNR_ATTN3_CHUNKS
python3 tools/parity_s1.py > $O/parity.txt 2>&1; tail -6 $O/parity.txt
