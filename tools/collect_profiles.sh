#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run through gpurun); summaries land in gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/final}; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python3 bench.py --streams 1 --opt chain_rows64=1 --opt gemm_wide_tiles=1 --steps 10 --warmup 3 --no-cpu-baseline --no-train > $O/kt.log 2>&1
python3 tools/rocpd_summary.py $(ls $O/kt/*.db $O/kt/*/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f -- python3 bench.py --streams 1 --opt chain_rows64=1 --opt gemm_wide_tiles=1 --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-train > $O/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w -- python3 bench.py --streams 1 --opt chain_rows64=1 --opt gemm_wide_tiles=1 --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-train > $O/pw.log 2>&1
python3 tools/pmc_traffic.py $(ls $O/pf/*.db $O/pf/*/*.db 2>/dev/null | head -1) $(ls $O/pw/*.db $O/pw/*/*.db 2>/dev/null | head -1) --json $O/hbm_traffic.json > $O/hbm_traffic_pmc.txt
rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python3 tools/bench_train.py 32 3 > $O/train.log 2>&1
python3 tools/rocpd_summary.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats.txt
rm -rf $O/kt $O/pf $O/pw $O/tr
tail -c 600 $O/bench.json; echo; head -12 $O/kernel_stats.txt | cut -c1-140; head -8 $O/hbm_traffic_pmc.txt | cut -c1-140; cat $O/hbm_traffic.json | head -12
