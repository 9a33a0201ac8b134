#!/bin/bash
# SQ counters of the S1 step per kernel (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02e}; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/counters_list.txt | sort -u | tr '\n' ' ' > $O/sq_counters.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_IFETCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P -d $O/p$i -o p -- python3 tools/s1_once.py > $O/p$i.log 2>&1
  python3 tools/rocpd_pmc.py $(ls $O/p$i/*.db $O/p$i/*/*.db 2>/dev/null | head -1) > $O/pmc_pass$i.txt 2>&1
  grep -E "kernel|panel_chain|attn3|gemm2" $O/pmc_pass$i.txt | cut -c1-260
  rm -rf $O/p$i
done
