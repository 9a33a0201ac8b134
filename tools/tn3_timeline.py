#!/usr/bin/env python3
"""Per-tile phases of gemm_tn3_kernel from a VNR_GEMM_TN3_TS dump: medians over waves, in kcyc of s_memtime."""
import struct, sys
import numpy as np
buf = open(sys.argv[1], 'rb').read()
off = 0
while off < len(buf):
    M, K, N, wgs = struct.unpack_from('4i', buf, off); off += 16
    n = wgs * 12 * 64
    ts = np.frombuffer(buf, dtype=np.uint64, count=n, offset=off).reshape(wgs, 12, 64).astype(np.int64); off += n * 8
    print("M=%d K=%d N=%d wgs=%d" % (M, K, N, wgs))
    for role, sel, names in (("multiplier waves", slice(0, 4), ["tile visible"] + ["products issued", "barrier passed"] * 40),
                             ("loader waves", slice(4, 12), ["prologue done"] + ["tile stored + refill issued", "barrier passed"] * 40)):
        part = ts[:, sel, :]
        w = part[part[:, :, 0] > 0]                    # [waves][64]
        if not len(w):
            continue
        rel = (w - w[:, :1]) / 1e3
        print(" %s (%d)" % (role, len(w)))
        prev = np.zeros(len(w))
        for i in range(1, 26):
            col = rel[:, i]
            if (w[:, i] == 0).all():
                break
            d = col - prev
            print("  %2d %-16s at %7.2f  (+%5.2f kcyc, p90 +%5.2f)" % (i, names[i - 1], np.median(col), np.median(d), np.percentile(d, 90)))
            prev = col
    life = (np.where(w > 0, w, 0).max(axis=1) - w[:, 0]) / 1e3          # (counters of different XCDs are not comparable: per wave only)
    print("  wave lifetime up to its last stamp: median %.1f  max %.1f kcyc" % (np.median(life), life.max()))
