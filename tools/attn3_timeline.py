#!/usr/bin/env python3
"""Decode the per-workgroup s_memtime stamps of attn3_kernel<true> (VNR_ATTN3_TS=<file>)."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read(); off = 0; last = None
while off < len(data):
    hdr = struct.unpack_from("4i", data, off); off += 16
    n = hdr[3]
    ts = np.frombuffer(data, dtype=np.uint64, count=n * 32, offset=off).reshape(n, 32).astype(np.int64); off += n * 256
    last = (hdr, ts)
hdr, ts = last
pairs, Tq, nchunk, n = hdr
tiles = (ts[:, 31] & 0xffffffff).astype(int); xcc = (ts[:, 31] >> 32).astype(int)
t0 = ts[:, 0].min()
print("pairs=%d Tq=%d chunks=%d wgs=%d; kernel span %.2f kcyc (first start .. last end)" % (pairs, Tq, nchunk, n, (ts[:, 30].max() - t0) / 1e3))
print("  start skew: median %.2f max %.2f kcyc" % (np.median(ts[:, 0] - t0) / 1e3, (ts[:, 0] - t0).max() / 1e3))
print("  operands landed (Q quarter, K, V): median %.2f  p90 %.2f kcyc after start" % (np.median(ts[:, 1] - ts[:, 0]) / 1e3, np.percentile(ts[:, 1] - ts[:, 0], 90) / 1e3))
names = ["S^T MFMAs done", "barrier 1", "next Q parked (vmcnt wait)", "alignment stores issued", "barrier 2", "context stores issued"]
for it in range(int(tiles.max())):
    sel = tiles > it
    b = 4 + 6 * it
    prev = ts[sel, 2] if it == 0 else ts[sel, b - 1]
    print("  tile %d (%d workgroups)" % (it, sel.sum()))
    for i, nm in enumerate(names):
        d = ts[sel, b + i] - prev
        print("    %-30s +%.2f (p90 %.2f) kcyc" % (nm, np.median(d) / 1e3, np.percentile(d, 90) / 1e3))
        prev = ts[sel, b + i]
life = ts[:, 30] - ts[:, 0]
print("  workgroup lifetime: median %.2f  p90 %.2f  max %.2f kcyc" % (np.median(life) / 1e3, np.percentile(life, 90) / 1e3, life.max() / 1e3))
print("  last workgroup end - median workgroup end: %.2f kcyc" % ((ts[:, 30].max() - np.median(ts[:, 30])) / 1e3))
