#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md prescribes) of the same command.  Writes the text table to stdout and, with --json, the per-launch
byte counts bench.py reports as `traffic`.

usage: pmc_traffic.py fetch.db write.db [--json profiles/rNN_hbm_traffic.json]
Units: KiB as reported by the counters; FETCH_SIZE is doubled (gfx950 wide-read correction of the guide), WRITE_SIZE is
uncalibrated."""
import collections
import json
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    acc = collections.OrderedDict()
    q = ("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? "
         "group by dispatch_id order by dispatch_id")
    for name, disp, val in c.execute(q, (counter,)):
        d = acc.setdefault(name.replace("void ", ""), [0, 0.0])
        d[0] += 1; d[1] += val
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py, per dispatch averages.")
    print("# Units: KiB as reported.  gfx950 correction (MI355X_MICROARCH.md section HBM): FETCH_SIZE under-reads wide coalesced")
    print("# streaming reads by 2x -> 'fetch_corr_MB' doubles it; WRITE_SIZE is uncalibrated.  The S1 working set largely fits")
    print("# the 256 MiB Infinity Cache, so these memory-side counters mostly see cache hits.")
    print("%-78s %7s %12s %14s %12s" % ("kernel", "calls", "fetch_KiB", "fetch_corr_MB", "write_KiB"))
    rows = {}
    for k, (n, v) in fetch.items():
        w = write.get(k, [1, 0.0])
        f_kib = v / max(n, 1); w_kib = w[1] / max(w[0], 1)
        rows[k] = (n, f_kib, w_kib)
        print("%-78s %7d %12.1f %14.2f %12.1f" % (k[:78], n, f_kib, 2 * f_kib * 1024 / 1e6, w_kib))
    if "--json" in sys.argv:
        def total(pred):
            n = sum(r[0] for k, r in rows.items() if pred(k))
            b = sum(r[0] * (2 * r[1] + r[2]) * 1024 for k, r in rows.items() if pred(k))
            return b / n if n else None
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import kernel_source_digest

        def gemm2_split(k):           # the SPLIT template argument of gemm2_kernel: 0 = fp32 MFMA, 1 / 2 = split-fp16 (true / false in round 1)
            if "gemm2_kernel<" not in k:
                return None
            targs = [x.strip() for x in k[k.index("<") + 1:k.index(">")].split(",")]
            split = targs[7] if len(targs) >= 9 else targs[-1]       # (round 3 appended the loader-wave count behind SPLIT)
            return split in ("1", "2", "true")
        out = {
            "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `bench.py --steps 2 --warmup 1 "
                      "--profile-steps 1 --in-flight 0 --no-train --no-cpu-baseline`; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                      "(gfx950 wide-read correction), WRITE_SIZE uncalibrated; KiB*1024",
            "kernel_source_digest": kernel_source_digest(),      # bench.py prints these figures only while the kernel sources are the same
            "chain_bytes_per_launch": total(lambda k: "panel_chain_kernel" in k or "panel_chain4_kernel" in k),
            "gemm_bytes_per_launch": total(lambda k: gemm2_split(k) is True),
            "gemm_fp32_bytes_per_launch": total(lambda k: gemm2_split(k) is False or "gemm_kernel" in k),
            "cross_attention_ali_bytes_per_launch": total(lambda k: "attn3_kernel<true>" in k or "attn2_kernel<true>" in k),
            "note": "memory-side (fabric) bytes; most of the S1 working set sits in the 256 MiB Infinity Cache, so these are "
                    "largely cache hits, not DRAM",
        }
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
