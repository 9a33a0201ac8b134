#!/usr/bin/env python3
"""round 5: stamp file of the 4-wave chain kernel (VNR_CHAIN_TS) grouped by the XCD a workgroup ran on (stamp slot 62 = HW_REG_XCC_ID):
s_memtime counts per XCD, so start spreads and launch spans only make sense inside one."""
import struct, sys
import numpy as np
data = open(sys.argv[1], "rb").read(); off = 0
want = int(sys.argv[2]) if len(sys.argv) > 2 else 14
n = 0
while off < len(data):
    M, Dw, ns, nw = struct.unpack_from("4i", data, off); off += 16
    ts = np.frombuffer(data, dtype=np.uint64, count=nw * 128, offset=off).reshape(nw, 128).astype(np.int64); off += nw * 1024
    if ns != want: continue
    pf = ts[ts[:, 63] >= 1000]; ts = ts[ts[:, 63] < 1000]
    n += 1
    if n > 2: break
    end = ts[:, 1 + 2 * ns]; life = (end - ts[:, 0]) / 1e3
    x = ts[:, 62]
    print("launch with %d stages, M=%d: lifetime median %.1f max %.1f" % (ns, M, np.median(life), life.max()))
    print("  xcd  wgs  start-spread  life-med  life-max  span   panel  " + " ".join("L%-2d  E%-2d " % (s, s) for s in range(ns)))
    for c in range(8):
        m = x == c
        if not m.any(): continue
        t = ts[m]
        row = "  %3d  %3d  %8.1f  %8.1f  %8.1f  %6.1f  %5.1f  " % (c, m.sum(), (t[:, 0].max() - t[:, 0].min()) / 1e3, np.median(life[m]), life[m].max(),
                                                                (end[m].max() - t[:, 0].min()) / 1e3, np.median(t[:, 1] - t[:, 0]) / 1e3)
        row += " ".join("%4.1f %4.1f" % (np.median(t[:, 2 + 2 * s] - t[:, 1 + 2 * s]) / 1e3, np.median(t[:, 3 + 2 * s] - t[:, 2 + 2 * s]) / 1e3) for s in range(ns))
        print(row)

    if len(pf) and ts[:, 96].any():
        # 100 MHz real-time stamps (the same clock everywhere), in us from the XCD's first worker's stage 0
        print("  stage starts of the workers (median of the XCD) against the prefetchers of the XCD (released / last line back: max over them), us:")
        for c in range(8):
            m = x == c; pm = pf[:, 62] == c
            if not m.any() or not pm.any(): continue
            t0 = ts[m, 96].min()
            wk = np.median(ts[m, 96:96 + ns] - t0, axis=0) / 100.0
            rel = (pf[pm, 32:32 + ns].max(axis=0) - t0) / 100.0
            don = (pf[pm, 96:96 + ns].max(axis=0) - t0) / 100.0
            print("   xcd %d workers    " % c + " ".join("%6.1f" % v for v in wk))
            print("         pf released" + " ".join("%6.1f" % v for v in rel))
            print("         pf done    " + " ".join("%6.1f" % v for v in don))
