#!/usr/bin/env python3
"""Decode the per-workgroup s_memtime stamps dumped by VNR_GEMM_TS=<file> (measurement aid)."""
import struct
import sys

import numpy as np


def main(path):
    data = open(path, "rb").read()
    off = 0
    seen = {}
    while off < len(data):
        M, N, K, BM, BN, NST, grid, ln = struct.unpack_from("8i", data, off); off += 32
        ts = np.frombuffer(data, dtype=np.uint64, count=grid * 8, offset=off).reshape(grid, 8).astype(np.int64); off += grid * 64
        seen[(M, N, K, BM, BN, NST, ln)] = ts      # keep the last (warm) launch of each shape
    for (M, N, K, BM, BN, NST, ln), ts in seen.items():
        # s_memtime ticks are shader cycles and the counter is per XCC: normalise entry times per XCC
        xcc = ts[:, 7].astype(int)
        t0 = np.array([ts[xcc == x, 0].min() if (xcc == x).any() else 0 for x in range(8)])[xcc]
        rel = (ts[:, :6] - t0[:, None]) / 1e3     # kilo-cycles
        ts = ts.copy(); ts[:, 2] = ts[:, 1]     # stamp 2 is no longer taken
        d = np.diff(ts[:, :6], axis=1) / 1e3
        print("M=%d N=%d K=%d tile %dx%d st%d ln=%d grid=%d" % (M, N, K, BM, BN, NST, ln, len(ts)))
        print("  kernel span (first entry -> last store done): %.2f kcyc" % rel[:, 5].max())
        print("  WG entry time   : min %.2f  median %.2f  p90 %.2f  max %.2f kcyc" % (rel[:, 0].min(), np.median(rel[:, 0]), np.percentile(rel[:, 0], 90), rel[:, 0].max()))
        names = ["prologue(addr+issue)", "(unused)", "main loop", "epilogue issue", "store drain"]
        for i, nm in enumerate(names):
            print("  %-26s: median %.2f  p90 %.2f  max %.2f kcyc" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90), d[:, i].max()))
        life = (ts[:, 5] - ts[:, 0]) / 1e3
        print("  WG lifetime              : median %.2f  p90 %.2f  max %.2f kcyc" % (np.median(life), np.percentile(life, 90), life.max()))
        print("  WGs per XCC:", np.bincount(xcc, minlength=8).tolist())


if __name__ == "__main__":
    main(sys.argv[1])
