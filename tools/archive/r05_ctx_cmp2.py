#!/usr/bin/env python3
"""round 5 debugging aid: VNR_DBG_CTX dumps of a fused launch against the matching unfused launches; min / max / mean / std of the difference
in the worst rows (a uniform, bounded difference = one wrong input element times a row of uniform weights)."""
import sys, struct
import numpy as np
def load(p):
    d = open(p, "rb").read(); off = 0; out = []
    while off < len(d):
        M, D, ns, w4 = struct.unpack_from("4i", d, off); off += 16
        out.append((ns, w4, np.frombuffer(d, np.float32, M * D, off).reshape(M, D))); off += M * D * 4
    return out
a, b = load(sys.argv[1]), load(sys.argv[2])
print("fused launches", [x[0] for x in a]); print("unfused launches", [x[0] for x in b])
fa = [x for x in a if np.abs(x[2]).max() > 0]; fb = [x for x in b if (x[0] == 2) == (len(sys.argv) <= 3) and np.abs(x[2]).max() > 0]
for (ns, _, ca), (nb, _, cb) in zip(fa, fb):
    d = ca - cb; rowmax = np.abs(d).max(axis=1)
    r = int(np.argmax(rowmax))
    w5 = np.argsort(-rowmax)[:5]
    for q in w5: print("    row %d: diff min %+.4e max %+.4e mean %+.4e std %.2e" % (q, d[q].min(), d[q].max(), d[q].mean(), d[q].std()))
    print("fused %d-stage vs unfused: max diff %.3e  median row-max %.3e  rows above 5x median %d;  worst row %d: diff min %.3e max %.3e mean %.3e (constant across columns?)" % (
        ns, rowmax.max(), np.median(rowmax), (rowmax > 5 * np.median(rowmax)).sum(), r, d[r].min(), d[r].max(), d[r].mean()))
