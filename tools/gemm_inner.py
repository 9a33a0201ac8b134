import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read(); off = 0; seen = {}
while off < len(data):
    M, N, K, BM, BN, NST, grid, ln = struct.unpack_from("8i", data, off); off += 32
    ts = np.frombuffer(data, dtype=np.uint64, count=grid * 16, offset=off).reshape(grid, 16).astype(np.int64); off += grid * 128
    seen[(M, N, K, BM, BN, NST, ln)] = ts
for k, ts in seen.items():
    if k[2] < 32 * 12: continue
    d = np.diff(ts[:, 8:14], axis=1)
    print(k, "grid", len(ts), " fragS1+mfmaS0 %.0f | waitcnt %.0f | barrier %.0f | mfmaS1+fragS0 %.0f | next-iter-same-point %.0f   (cycles, median)" % tuple(np.median(d, axis=0)))
