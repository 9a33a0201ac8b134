#!/bin/bash
# Round 6: collect the round's profile artefacts on the GPU box (run through gpurun); summaries land in gpurun_out/final6/.
# The per-kernel blocks of bench.py and these rocprofv3 runs describe the SAME schedule: one batch in flight, default engine options.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the profiler's preloaded library initialises HIP before Python starts: the engine's own setdefault (vaenar_tts_amd/_lib.py) would come too
# late, and the profiled runs would map streams onto 4 hardware queues while bench.py measures with 8
export GPU_MAX_HW_QUEUES=8
O=${1:-gpurun_out/final6}; mkdir -p $O
R=r06
P="--in-flight 0 --no-cpu-baseline --no-train --no-exact-pass --no-attn-phase"
# HBM traffic: separate --pmc passes (kernel trace only), FETCH_SIZE doubled per MI355X_MICROARCH.md
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 $P > $O/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w -- python3 bench.py --steps 2 --warmup 1 --profile-steps 1 $P > $O/pw.log 2>&1
python3 tools/pmc_traffic.py $(ls $O/pf/*.db $O/pf/*/*.db 2>/dev/null | head -1) $(ls $O/pw/*.db $O/pw/*/*.db 2>/dev/null | head -1) --json profiles/${R}_hbm_traffic.json > $O/hbm_traffic_pmc.txt
cp profiles/${R}_hbm_traffic.json $O/hbm_traffic.json
python3 bench.py > $O/bench.out 2> $O/bench.err; grep "^{" $O/bench.out | tail -1 > $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python3 bench.py --steps 10 --warmup 3 $P > $O/kt.log 2>&1
python3 tools/rocpd_summary.py $(ls $O/kt/*.db $O/kt/*/*.db 2>/dev/null | head -1) > $O/kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python3 tools/bench_train.py 32 3 > $O/train.log 2>&1
python3 tools/rocpd_summary.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) > $O/train_kernel_stats.txt
python3 tools/train_timeline.py $(ls $O/tr/*.db $O/tr/*/*.db 2>/dev/null | head -1) 4 > $O/train_timeline.txt
rm -rf $O/kt $O/pf $O/pw $O/tr
# SQ counters of the S1 step per kernel (three separate --pmc passes, kernel trace only): the round-6 counter pass of the chain kernel
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_IFETCH"
i=0; : > $O/chain_pmc.txt
for PP in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PP -d $O/p$i -o p -- python3 tools/s1_once.py > $O/p$i.log 2>&1
  python3 tools/rocpd_pmc.py $(ls $O/p$i/*.db $O/p$i/*/*.db 2>/dev/null | head -1) >> $O/chain_pmc.txt 2>&1
  echo >> $O/chain_pmc.txt
  rm -rf $O/p$i
done
python3 tools/trim_chain_pmc.py $O/chain_pmc.txt $(python3 -c 'import bench; print(bench.kernel_source_digest())') > $O/chain_pmc_trimmed.txt
rm -f /tmp/cts.bin; VNR_CHAIN_TS=/tmp/cts.bin VNR_CHAIN_TS_STAGE=1 python3 tools/s1_once.py > /dev/null 2>&1; python3 tools/chain_timeline.py /tmp/cts.bin > $O/chain_rows32_timeline.txt 2>&1
python3 tools/chain_cycle_sums.py /tmp/cts.bin > $O/chain_cycle_sums.txt 2>&1
rm -f /tmp/cts8.bin; VNR_CHAIN_WAVES4=0 VNR_CHAIN_TS=/tmp/cts8.bin VNR_CHAIN_TS_STAGE=1 python3 tools/s1_once.py > /dev/null 2>&1; python3 tools/chain_timeline.py /tmp/cts8.bin > $O/chain_timeline_waves8.txt 2>&1
echo "--- chain_waves4 = 0 (the 8-wave kernel) on the same box:" >> $O/chain_cycle_sums.txt; python3 tools/chain_cycle_sums.py /tmp/cts8.bin >> $O/chain_cycle_sums.txt 2>&1
rm -f /tmp/g.ts; VNR_GEMM_TS=/tmp/g.ts python3 tools/s1_once.py > /dev/null 2>&1; python3 tools/gemm_timeline.py /tmp/g.ts > $O/gemm_timeline.txt 2>&1
rocprofv3 --kernel-trace -d $O/sq -o s -- python3 bench.py --steps 6 --warmup 3 --profile-steps 0 $P > $O/sq.log 2>&1; python3 tools/launch_sequence.py $(ls $O/sq/*.db $O/sq/*/*.db 2>/dev/null | head -1) > $O/launch_sequence.txt; rm -rf $O/sq
tail -c 600 $O/bench.json; echo; head -14 $O/kernel_stats.txt | cut -c1-140; head -8 $O/hbm_traffic_pmc.txt | cut -c1-140; cat $O/hbm_traffic.json | head -14
