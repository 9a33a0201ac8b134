#!/bin/bash
# round 6: copy the summaries of one tools/collect_profiles_r06.sh run (gpurun_out/<dir>) into profiles/r06_*; usage: r06_adopt_collection.sh gpurun_out/final6c [bench json to use]
O=$1; B=${2:-$O/bench.json}
cp $B profiles/r06_bench.json
cp $O/kernel_stats.txt profiles/r06_kernel_stats.txt; cp $O/launch_sequence.txt profiles/r06_launch_sequence.txt
cp $O/hbm_traffic.json profiles/r06_hbm_traffic.json; cp $O/hbm_traffic_pmc.txt profiles/r06_hbm_traffic_pmc.txt
cp $O/chain_pmc_trimmed.txt profiles/r06_chain_pmc.txt
cp $O/chain_rows32_timeline.txt profiles/r06_chain_rows32_timeline.txt; cp $O/chain_timeline_waves8.txt profiles/r06_chain_timeline_waves8.txt
cp $O/chain_cycle_sums.txt profiles/r06_chain_cycle_sums.txt; cp $O/gemm_timeline.txt profiles/r06_gemm_timeline.txt
cp $O/train_kernel_stats.txt profiles/r06_train_kernel_stats.txt; cp $O/train_timeline.txt profiles/r06_train_timeline.txt
python3 - <<'PY'
import json
d = json.load(open("profiles/r06_bench.json")); t = json.load(open("profiles/r06_hbm_traffic.json"))
print("bench digest", d["kernel_source_digest"], "traffic digest", t.get("kernel_source_digest"), "traffic in line:", d["roofline"]["traffic"])
print("ms %.3f  8-wave %.3f  exact %.2f  frac %.3f  chain %.3f  train %.2f / %.2f  parity %.2e" % (d["ms_per_step"], d["chain_kernel_8wave"]["ms_per_step"], d["exact_fp32"]["ms_per_step"],
      d["roofline"]["frac"], d["end_to_end"]["kernel_ms_per_step"]["chain"] + d["end_to_end"]["kernel_ms_per_step"]["chain_ali"], d["training"]["ms_per_step"], d["training"]["atomic_mode_ms_per_step"], d["parity"]["max_abs_mel_err"]))
PY
