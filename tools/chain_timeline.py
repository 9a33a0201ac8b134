#!/usr/bin/env python3
"""Decode per-stage s_memtime stamps of the row-panel chain kernel (VNR_CHAIN_TS=<file>)."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read(); off = 0; seen = {}
while off < len(data):
    M, D, ns, nw = struct.unpack_from("4i", data, off); off += 16
    ts = np.frombuffer(data, dtype=np.uint64, count=nw * 64, offset=off).reshape(nw, 64).astype(np.int64); off += nw * 512
    seen[(M, D, ns)] = ts
for (M, D, ns), ts in seen.items():
    print("M=%d D=%d stages=%d wgs=%d   lifetime median %.1f kcyc" % (M, D, ns, len(ts), np.median(ts[:, 1 + 2 * ns] - ts[:, 0]) / 1e3))
    print("  panel load: %.2f kcyc" % (np.median(ts[:, 1] - ts[:, 0]) / 1e3))
    for s in range(ns):
        loop = np.median(ts[:, 2 + 2 * s] - ts[:, 1 + 2 * s]) / 1e3
        epi = np.median(ts[:, 3 + 2 * s] - ts[:, 2 + 2 * s]) / 1e3
        print("  stage %2d: loop %.2f  epilogue %.2f kcyc" % (s, loop, epi))
    if ns >= 2:
        w = ts[:, 32:64].reshape(len(ts), 8, 4)
        base = w[:, :, 0].min(axis=1, keepdims=True)
        for i, name in enumerate(("loop start", "loop end", "after barrier 1", "stage end")):
            print("  stage 1 per wave %-16s" % name, " ".join("%6.2f" % x for x in np.median(w[:, :, i] - base, axis=0) / 1e3))
