#!/bin/bash
# round 5: A/B of environment / option settings on ONE box.  usage: r05_ab.sh "<label>|<env assignments>|<bench args>" ...
# prints ms per S1 step and the kernel-class times (dispatch events) of each setting, alternating twice
mkdir -p gpurun_out/r05
for rep in $(seq 1 ${REPS:-2}); do
for cfg in "$@"; do
  IFS='|' read -r label envs bargs <<< "$cfg"
  line=$(env $envs python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train --no-exact-pass --in-flight 0 --no-attn-phase $bargs 2>&1 | grep '^{' | tail -1)
  echo "$line" > gpurun_out/r05/ab_${label}_$rep.json
  python - "$label" "$rep" <<PY
import json, sys
d = json.loads(open("gpurun_out/r05/ab_%s_%s.json" % (sys.argv[1], sys.argv[2])).read())
k = d["end_to_end"]["kernel_ms_per_step"]
k["chain"] = k.get("chain", 0) + k.get("chain_ali", 0)
ca = d.get("roofline_cross_attention", {}).get("avg_launch_us", -1)
print("%-14s rep %s  ms %.3f  launches %d  chain %.3f  gemm %.3f  self %.3f  ali_us %.2f  ln %.3f misc %.3f  h2h_med %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"],
      d["end_to_end"]["kernel_launches_per_step"], k["chain"], k["gemm"], k["attn_self"], ca, k["layer_norm"], k["misc"], d["latency_host_to_host_ms"]["median"]))
PY
done; done
