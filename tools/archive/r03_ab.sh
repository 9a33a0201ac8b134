#!/bin/bash
# round 3: A/B of environment / option settings on ONE box.  usage: r03_ab.sh "<label>|<env assignments>|<bench args>" ...
# prints ms per S1 step and the kernel-class times (dispatch events) of each setting, alternating twice
mkdir -p gpurun_out/r03
for rep in 1 2; do
for cfg in "$@"; do
  IFS='|' read -r label envs bargs <<< "$cfg"
  line=$(env $envs python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train --in-flight 0 $bargs 2>&1 | grep '^{' | tail -1)
  echo "$line" > gpurun_out/r03/ab_${label}_$rep.json
  python - "$label" "$rep" <<PY
import json, sys
d = json.loads(open("gpurun_out/r03/ab_%s_%s.json" % (sys.argv[1], sys.argv[2])).read())
k = d["end_to_end"]["kernel_ms_per_step"]
print("%-14s rep %s  ms %.3f  chain %.3f  gemm %.3f  self %.3f  ali_us %.2f  err %.2e" % (sys.argv[1], sys.argv[2], d["ms_per_step"], k["chain"], k["gemm"], k["attn_self"],
      d["roofline_cross_attention"]["avg_launch_us"], d["parity"]["max_abs_mel_err"] if "parity" in d else -1))
PY
done; done
