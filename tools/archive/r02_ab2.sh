#!/bin/bash
# quick check of a kernel change: core GPU tests + single-stream bench breakdown (+ optional env A/B list in $AB: "NAME=VAL ...")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02d}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_golden.py -m gpu -x -q > $O/pytest_core.log 2>&1; echo "rc=$?" >> $O/pytest_core.log; tail -3 $O/pytest_core.log
B="python3 bench.py --steps 20 --warmup 3 --in-flight 0 --no-cpu-baseline --no-train"
show() { python3 - $1 "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["end_to_end"]["kernel_ms_per_step"]
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "chain %.3f gemm %.3f self %.3f cross %.3f ln %.3f misc %.3f ali_us %.2f" % (k["chain"],k["gemm"],k["attn_self"],k["attn_cross"],k["layer_norm"],k["misc"], d["roofline_cross_attention"]["avg_launch_us"]), "frac_x %.3f frac_dom %.3f" % (d["roofline_cross_attention"]["frac"], d["roofline"]["frac"]))
PY
}
$B > $O/bench_base.json 2> $O/bench_base.err; show $O/bench_base.json base
i=0
for kv in $AB; do
  i=$((i+1))
  env $kv $B > $O/bench_ab$i.json 2> $O/bench_ab$i.err; show $O/bench_ab$i.json "$kv"
done
