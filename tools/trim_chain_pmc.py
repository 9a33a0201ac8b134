#!/usr/bin/env python3
"""Condense the three rocprofv3 --pmc passes of tools/s1_once.py (tools/rocpd_pmc.py output, one block per pass) into the committed
summary: the attention / GEMM / chain kernels only (weight-preparation kernels of vnr_finalize_weights dropped), the chain kernel's
counter means over its launches and its ratios against SQ_WAVE_CYCLES.  usage: trim_chain_pmc.py <chain_pmc.txt> <digest> > profiles/rNN_chain_pmc.txt"""
import statistics
import sys

src = open(sys.argv[1]).read().split('\n\n')
digest = sys.argv[2] if len(sys.argv) > 2 else "?"
out = ["# SQ counters of the S1 inference step per kernel (tools/collect_profiles_r06.sh: three separate `rocprofv3 --kernel-trace --pmc ...`",
       "# passes of tools/s1_once.py, averages per dispatch; weight-preparation kernels of vnr_finalize_weights left out).  Kernel sources %s." % digest,
       "# panel_chain4_kernel (panel_chain_kernel<1> before the 4-wave kernel became the default): 15 launches per step (13 block launches with everything of a block behind its self-attention, the coupling and",
       "# the next pre-chain, + the first pre-chain): see the ratios at the end."]
keep = ("kernel ", "vnr::panel_chain", "vnr::attn3", "vnr::gemm2_kernel", "vnr::layer_norm", "vnr::gather_rows", "panel_chain", "attn3", "gemm2_kernel", "layer_norm", "gather_rows")
vals = {}
for blk in src:
    lines = [l for l in blk.split('\n') if l.strip()]
    if not lines:
        continue
    out.append("")
    out += [l[:260] for l in lines if l.startswith(keep)]
    names = lines[0].split()[3:]
    for l in lines:
        if "panel_chain4_kernel" in l.split("(")[0] or "panel_chain_kernel<1>" in l.split("(")[0]:
            nums = l.split(")")[-1].split()
            for n, v in zip(names, nums[2:]):
                vals.setdefault(n, []).append(float(v))
out += ["", "# the chain kernel, mean over its launches of one step (counter names as rocprofv3 prints them, truncated on the left):"]
out += ["#   %-18s %.4g" % (n, statistics.mean(v)) for n, v in vals.items()]


def g(k):
    for n, v in vals.items():
        if n.endswith(k):
            return statistics.mean(v)
    return None


wc, mf, wa, ai, wi = g("SQ_WAVE_CYCLES"), g("MA_BUSY_CYCLES"), g("SQ_WAIT_ANY"), g("CTIVE_INST_ANY"), g("_WAIT_INST_ANY")
if wc and mf and wa and ai and wi:
    out.append("# ratios: MFMA pipe busy / wave cycles = %.3f ; SQ_WAIT_ANY / wave cycles = %.3f ; SQ_WAIT_INST_ANY / wave cycles = %.3f ; "
               "SQ_ACTIVE_INST_ANY / wave cycles = %.3f" % (mf / wc, wa / wc, wi / wc, ai / wc))
print("\n".join(out))
