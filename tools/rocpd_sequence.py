#!/usr/bin/env python3
"""Print the kernel sequence of the LAST `n` dispatches of a rocprofv3 rocpd database with durations and the idle gap
before each kernel (start - previous end).  usage: rocpd_sequence.py file.db [n]"""
import sqlite3
import sys


def main(path, n=140):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, workgroup_x*workgroup_y*workgroup_z, grid_x*grid_y*grid_z from kernels order by start").fetchall()
    rows = rows[-n:]
    prev = None
    busy = gap = 0.0
    for name, s, e, wg, grid in rows:
        g = (s - prev) / 1e3 if prev is not None else 0.0
        d = (e - s) / 1e3
        busy += d
        gap += max(g, 0.0)
        short = name.replace("void ", "").replace("vnr::", "")[:70]
        print("%8.2f us  gap %6.2f  wgs %6d  %s" % (d, g, grid // max(wg, 1), short))
        prev = e
    print("# busy %.1f us, gaps %.1f us, span %.1f us" % (busy, gap, (rows[-1][2] - rows[0][1]) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 140)
