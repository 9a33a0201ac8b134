#!/usr/bin/env python3
"""Per-XCC start/end skew of attention2 workgroups from the VNR_ATTN_TS stamps (every record in the file)."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read(); off = 0; recs = []
while off < len(data):
    hdr = struct.unpack_from("8i", data, off); off += 32
    n = hdr[6]
    ts = np.frombuffer(data, dtype=np.uint64, count=n * 8, offset=off).reshape(n, 8).astype(np.int64); off += n * 64
    recs.append((hdr, ts))
for hdr, ts in recs[-int(sys.argv[2]) if len(sys.argv) > 2 else 0:]:
    B, H, Tq, Tk, causal, ali, n, nqb = hdr
    xcc = ts[:, 7].astype(int)
    print("B=%d H=%d Tq=%d Tk=%d causal=%d ali=%d wgs=%d" % (B, H, Tq, Tk, causal, ali, n))
    for x in range(8):
        s = ts[xcc == x]
        if not len(s): continue
        t0 = s[:, 0].min()
        st = np.sort(s[:, 0] - t0); en = s[:, 4] - t0
        print("  xcc %d: %3d WGs; start p50 %6d p90 %6d max %6d | end min %6d p50 %6d max %6d | life p50 %6d" % (
            x, len(s), np.median(st), np.percentile(st, 90), st.max(), en.min(), np.median(en), en.max(), np.median(s[:, 4] - s[:, 0])))
