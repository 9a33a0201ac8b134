#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run through gpurun); summaries land in gpurun_out/final/.
# The per-kernel blocks of bench.py and these rocprofv3 runs describe the SAME schedule: one batch in flight, default engine options.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/final}; mkdir -p $O
R=${ROUND:-r03}            # the PMC traffic record is written as profiles/${R}_hbm_traffic.json on the box and copied to $O
P="--in-flight 0 --no-cpu-baseline --no-train"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 $P > $O/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 $P > $O/pw.log 2>&1
python3 tools/pmc_traffic.py $(ls $O/pf/*.db $O/pf/*/*.db 2>/dev/null | head -1) $(ls $O/pw/*.db $O/pw/*/*.db 2>/dev/null | head -1) --json profiles/${R}_hbm_traffic.json > $O/hbm_traffic_pmc.txt
cp profiles/${R}_hbm_traffic.json $O/hbm_traffic.json
python3 bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python3 bench.py --steps 10 --warmup 3 $P > $O/kt.log 2>&1
python3 tools/rocpd_summary.py $(ls $O/kt/*.db $O/kt/*/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python3 tools/bench_train.py 32 3 > $O/train.log 2>&1
python3 tools/rocpd_summary.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats.txt
python3 tools/train_timeline.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) 4 > $O/train_timeline.txt
rm -rf $O/kt $O/pf $O/pw $O/tr
for c in 8; do rm -f /tmp/a3.bin; VNR_ATTN3_TS=/tmp/a3.bin python3 tools/s1_once.py > /dev/null 2>&1; python3 tools/attn3_timeline.py /tmp/a3.bin > $O/attn3_timeline.txt 2>&1; done
rm -f /tmp/cts.bin; VNR_CHAIN_TS=/tmp/cts.bin VNR_CHAIN_TS_STAGE=1 python3 tools/s1_once.py > /dev/null 2>&1; python3 tools/chain_timeline.py /tmp/cts.bin > $O/chain_rows32_timeline.txt 2>&1
rm -f /tmp/tn3.bin; TN_SHAPES=25600x512x512,12800x256x256 VNR_GEMM_TN3_TS=/tmp/tn3.bin python3 tools/tn_bench.py 1 > /dev/null 2>&1; python3 tools/tn3_timeline.py /tmp/tn3.bin > $O/tn3_timeline.txt 2>&1
rm -f /tmp/g.ts; VNR_GEMM_TS=/tmp/g.ts python3 tools/s1_once.py > /dev/null 2>&1; python3 tools/gemm_timeline.py /tmp/g.ts > $O/gemm_timeline.txt 2>&1
rocprofv3 --kernel-trace -d $O/sq -o s -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 --in-flight 0 --no-cpu-baseline --no-train > $O/sq.log 2>&1; python3 tools/launch_sequence.py $(ls $O/sq/*.db $O/sq/*/*.db 2>/dev/null | head -1) > $O/launch_sequence.txt; rm -rf $O/sq
tail -c 800 $O/bench.json; echo; head -14 $O/kernel_stats.txt | cut -c1-140; head -8 $O/hbm_traffic_pmc.txt | cut -c1-140; cat $O/hbm_traffic.json | head -12
