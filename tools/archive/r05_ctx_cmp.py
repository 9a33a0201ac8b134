#!/usr/bin/env python3
"""round 5 debugging aid: compare two VNR_DBG_CTX dump files (see tools/r05_ctx_dump.py) row by row and head by head."""
import sys, struct
import numpy as np
def load(p):
    d = open(p, "rb").read(); off = 0; out = []
    while off < len(d):
        M, D, ns, w4 = struct.unpack_from("4i", d, off); off += 16
        out.append((ns, w4, np.frombuffer(d, np.float32, M * D, off).reshape(M, D))); off += M * D * 4
    return out
a, b = load(sys.argv[1]), load(sys.argv[2])
print("launches", [(x[0], x[1]) for x in a], [(x[0], x[1]) for x in b])
for (ns, wa, ca), (_, wb, cb) in zip(a[-2:], b[-2:]):
    d = np.abs(ca - cb); rowmax = d.max(axis=1)
    print("launch with %d stages: ctx max|a| %.3f  max diff %.3e  median row-max diff %.3e; rows above 5x the median: %d" % (ns, np.abs(ca).max(), d.max(), np.median(rowmax), (rowmax > 5 * np.median(rowmax)).sum()))
    worst = np.argsort(-rowmax)[:6]
    for r in worst:
        heads = [float(d[r, 64 * h:64 * h + 64].max()) for h in range(4)]
        print("   row %5d: max diff %.3e; per head %s; |ctx| row max %.3f" % (r, rowmax[r], " ".join("%.1e" % x for x in heads), np.abs(cb[r]).max()))
