#!/usr/bin/env python3
"""Decode per-stage s_memtime stamps of the row-panel chain kernel (VNR_CHAIN_TS=<file>, VNR_CHAIN_TS_STAGE=<stage with per-wave stamps>)."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read(); off = 0; seen = {}
while off < len(data):
    M, Dw, ns, nw = struct.unpack_from("4i", data, off); off += 16
    ts = np.frombuffer(data, dtype=np.uint64, count=nw * 128, offset=off).reshape(nw, 128).astype(np.int64); off += nw * 1024
    seen[(M, Dw, ns)] = ts[ts[:, 63] < 1000]               # (rows of prefetch workgroups carry 1000 + rank in slot 63: tools/archive/r05_ts_xcd.py reads them)
for (M, Dw, ns), ts in seen.items():
    D, ali, att = Dw & 0xffff, (Dw >> 16) & 1, Dw >> 20        # (round 5: the header's D word carries the fused-attention flags)
    print("M=%d D=%d stages=%d wgs=%d%s   lifetime median %.1f kcyc" % (M, D, ns, len(ts), (" attention in front of stage %d%s" % (att, " + alignments" if ali else "")) if att else "",
                                                                       np.median(ts[:, 1 + 2 * ns] - ts[:, 0]) / 1e3))
    life = (ts[:, 1 + 2 * ns] - ts[:, 0]) / 1e3
    print("  lifetime p10 %.1f  p90 %.1f  max %.1f kcyc" % (np.percentile(life, 10), np.percentile(life, 90), life.max()))
    # s_memtime counts per XCD / shader engine (different bases): lifetimes compare, start and end times across workgroups do not
    xs = [np.arange(len(ts)) % 8 == x for x in range(8)]
    print("  by XCD (workgroup mod 8): median lifetime " + " ".join("%6.1f" % np.median(life[m]) for m in xs))
    if att and ts[:, 61].any():
        print("  attention phase: %.2f kcyc (inside stage %d's \"loop\" figure below)" % (np.median(ts[:, 61] - ts[:, 60]) / 1e3, att))
    print("  panel load: %.2f kcyc" % (np.median(ts[:, 1] - ts[:, 0]) / 1e3))
    if ts[:, 56].any():                                       # (4-wave kernel: sub-phases)
        print("    row reads issued +%.2f, weight head k-tiles 0-3 issued +%.2f, zero fill + rows converted +%.2f, k-tiles 4-7 + parameter DMA issued and landed +%.2f kcyc" % tuple(
            np.median(ts[:, b] - ts[:, a]) / 1e3 for a, b in ((0, 56), (56, 59), (59, 58), (58, 1))))
    tl = te = 0.0
    for s in range(ns):
        loop = np.median(ts[:, 2 + 2 * s] - ts[:, 1 + 2 * s]) / 1e3
        epi = np.median(ts[:, 3 + 2 * s] - ts[:, 2 + 2 * s]) / 1e3
        tl += loop; te += epi
        print("  stage %2d: loop %.2f  epilogue %.2f kcyc" % (s, loop, epi))
    print("  sum of loops %.1f, of epilogues %.1f kcyc" % (tl, te))
    w = ts[:, 64:128].reshape(len(ts), 8, 8)
    if ts[:, 56].any(): w = w[:, :4, :]                      # (the 4-wave kernel: slots 96.. carry real-time stamps, not waves 4-7)
    if w[:, :, 0].any():
        nzw = w[:, :, 0].any(axis=0)                          # (the 4-wave kernel stamps four waves)
        w = w[:, nzw, :]
        base = w[:, :, 0].min(axis=1, keepdims=True)
        for i, name in ((0, "stage start"), (1, "loop end"), (4, "values ready"), (6, "residual / PE added"), (7, "row statistics ready"), (2, "after LN / in-place barrier"), (5, "outputs issued"), (3, "stage end (barrier)")):
            if not w[:, :, i].any(): continue                  # (a stamp this stage's path does not take)
            print("  per wave %-28s" % name, " ".join("%6.2f" % x for x in np.median(w[:, :, i] - base, axis=0) / 1e3))
