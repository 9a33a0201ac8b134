#!/usr/bin/env python3
"""Per-shape kernel time of tools/tn_bench.py under rocprofv3: the kernel-gradient GEMM dispatches in order, `reps` per shape."""
import glob, sqlite3, sys, os
d = sys.argv[1]
db = (glob.glob(d + '/*.db') + glob.glob(d + '/*/*.db'))[0]
c = sqlite3.connect(db)
rows = c.execute("select name, end - start, grid_x from kernels where name like '%gemm_tn%' order by start").fetchall()
reps = 5
for i in range(0, len(rows), reps):
    blk = rows[i:i + reps]
    print("  %-28s wgs %5d  avg %.2f us  min %.2f us" % (blk[0][0].split('(')[0].replace('vnr::', '').replace('void ', ''), blk[0][2] // 256,
                                                      sum(r[1] for r in blk) / len(blk) / 1e3, min(r[1] for r in blk) / 1e3))
