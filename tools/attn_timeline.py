#!/usr/bin/env python3
"""Decode per-workgroup s_memtime stamps of attention2 (VNR_ATTN_TS=<file>)."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read(); off = 0; seen = {}
while off < len(data):
    hdr = struct.unpack_from("8i", data, off); off += 32
    n = hdr[6]
    ts = np.frombuffer(data, dtype=np.uint64, count=n * 8, offset=off).reshape(n, 8).astype(np.int64); off += n * 64
    seen[hdr] = ts
for hdr, ts in seen.items():
    B, H, Tq, Tk, causal, ali, n, nqb = hdr
    d = np.diff(ts[:, :5], axis=1) / 1e3
    life = (ts[:, 4] - ts[:, 0]) / 1e3
    print("B=%d H=%d Tq=%d Tk=%d causal=%d ali=%d wgs=%d" % (B, H, Tq, Tk, causal, ali, n))
    for i, nm in enumerate(["prologue (Q loads, DMA issue)", "tile loop", "merge + store issue", "store drain"]):
        print("  %-30s median %.2f  p90 %.2f  max %.2f kcyc" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
    print("  %-30s median %.2f  p90 %.2f  max %.2f kcyc" % ("WG lifetime", np.median(life), np.percentile(life, 90), life.max()))
    for t in sorted(set(ts[:, 5])):
        sel = ts[:, 5] == t
        print("    tiles=%d: %d WGs, loop median %.2f kcyc, lifetime median %.2f" % (t, sel.sum(), np.median(d[sel, 1]), np.median(life[sel])))
    xcc = ts[:, 7].astype(int)
    span = max((ts[xcc == x, 4].max() - ts[xcc == x, 0].min()) for x in range(8) if (xcc == x).any()) / 1e3
    print("  kernel span on the busiest XCC: %.2f kcyc" % span)
