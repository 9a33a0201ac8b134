#!/bin/bash
# copy the summaries of tools/collect_profiles.sh (gpurun_out/<dir>) into the tracked profiles/ set of the round
O=${1:-gpurun_out/final3}; R=${ROUND:-r03}
for f in bench.json kernel_stats.txt hbm_traffic.json hbm_traffic_pmc.txt train_kernel_stats.txt train_timeline.txt gemm_timeline.txt chain_rows32_timeline.txt attn3_timeline.txt launch_sequence.txt tn3_timeline.txt; do
  cp $O/$f profiles/${R}_$f
done
