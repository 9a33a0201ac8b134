#!/bin/bash
# round 4: SQ counters and fabric traffic of the T1 training step's heavy kernels (separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04pmc; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
: > $O/train_pmc.txt
i=0
for PP in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  VNR_TRAIN_ONE_STREAM=1 rocprofv3 --kernel-trace --pmc $PP -d $O/p$i -o p -- python3 tools/bench_train.py 32 1 > $O/p$i.log 2>&1
  python3 tools/rocpd_pmc.py $(ls $O/p$i/*.db $O/p$i/*/*.db 2>/dev/null | head -1) >> $O/train_pmc.txt 2>&1
  echo >> $O/train_pmc.txt
  rm -rf $O/p$i
done
python3 - <<PY
import re
src = open("$O/train_pmc.txt").read().split("\n\n")
keep = ("kernel ", "vnr::gemm_tn3_kernel", "vnr::bwd_chain_kernel<2>", "vnr::panel_chain_kernel<2>", "vnr::attn_bwd_dq3", "vnr::attn_bwd_dkv3", "vnr::attn2_kernel<true, 7>", "vnr::gemm2_kernel<64, 64, 2, 2, 3, 0, false, 1, 0>", "vnr::gemm_tn_split")
out = ["# SQ counters (two passes) and fabric traffic (FETCH_SIZE / WRITE_SIZE in KiB as reported; FETCH_SIZE to be doubled per MI355X_MICROARCH.md) of the heavy kernels of the",
       "# T1 training step (B = 32, rf = 2 and 5, one step each), VNR_TRAIN_ONE_STREAM=1 so that every kernel runs alone: averages per dispatch and grid size."]
for blk in src:
    lines = [l for l in blk.split("\n") if l.strip()]
    if not lines: continue
    out.append("")
    out += [l[:250] for l in lines if l.startswith(keep)]
open("$O/train_pmc_trimmed.txt", "w").write("\n".join(out) + "\n")
PY
wc -l $O/train_pmc_trimmed.txt; grep "gemm_tn3" $O/train_pmc_trimmed.txt | head -8 | cut -c1-230
