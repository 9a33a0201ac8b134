#!/usr/bin/env python3
"""round 5: one number per stamp file -- the sum over the chain launches of one inference of the median workgroup lifetime (kcyc), and the
same per launch type.  Box clocks differ by a few per cent; cycle counts of two kernels on ONE box compare cleanly."""
import struct, sys, collections
import numpy as np
data = open(sys.argv[1], "rb").read(); off = 0; c = collections.OrderedDict()
while off < len(data):
    M, Dw, ns, nw = struct.unpack_from("4i", data, off); off += 16
    ts = np.frombuffer(data, dtype=np.uint64, count=nw * 128, offset=off).reshape(nw, 128).astype(np.int64); off += nw * 1024
    w = ts[(ts[:, 63] < 1000) & (ts[:, 0] != 0) & (ts[:, 1 + 2 * ns] != 0)]
    if not len(w): continue
    life = (w[:, 1 + 2 * ns] - w[:, 0]) / 1e3
    c.setdefault((Dw & 0xffff, ns, (Dw >> 16) & 1), []).append((np.median(life), life.max()))
tot = sum(m for v in c.values() for m, _ in v); totmax = sum(x for v in c.values() for _, x in v)
print("launches %d  sum of median lifetimes %.0f kcyc  sum of slowest-workgroup lifetimes %.0f kcyc" % (sum(len(v) for v in c.values()), tot, totmax))
for k, v in c.items():
    print("  D=%d stages=%d%s  x%d  median %.1f  slowest %.1f" % (k[0], k[1], " +ali" if k[2] else "", len(v), np.median([m for m, _ in v]), np.median([x for _, x in v])))
