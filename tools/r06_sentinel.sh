#!/bin/bash
# Round 6: the range sentinel on the box -- scalar-store probe, the range tests, and a same-box A/B of the S1 step with the probes on / off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r6sent; mkdir -p $O
./tools/probes/bin/sstore_probe > $O/sstore_probe.txt 2>&1; cat $O/sstore_probe.txt
timeout 1500 python -m pytest tests/test_gpu_range.py -x -q -s 2>&1 | tail -60 > $O/range_tests.txt; tail -40 $O/range_tests.txt
P="--in-flight 0 --no-cpu-baseline --no-train --no-exact-pass --no-attn-phase"
for rep in 1 2; do
for v in 1 0; do
  python3 bench.py --steps 20 --warmup 5 $P --opt range_sentinel=$v 2>/dev/null | grep "^{" | tail -1 > $O/ab_${v}_${rep}.json
  python3 - $O/ab_${v}_${rep}.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); k=d["end_to_end"]["kernel_ms_per_step"]
print("sentinel=%s ms %.4f chain %.4f gemm %.4f self %.4f h2h %.4f" % (sys.argv[2], d["ms_per_step"], k["chain"]+k.get("chain_ali",0), k["gemm"], k["attn_self"], d["latency_host_to_host_ms"]["median"]))
PY
done; done 2>&1 | tee $O/ab.txt
