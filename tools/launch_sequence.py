#!/usr/bin/env python3
"""The launch sequence of the LAST S1 inference step in a rocprofv3 (rocpd sqlite) kernel trace: one line per dispatch with its
start offset, duration and the idle gap in front of it.  A step starts at the text-embedding gather (one per step).

usage: python tools/launch_sequence.py trace_results.db > profiles/rNN_launch_sequence.txt"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    first = max(i for i, r in enumerate(rows) if "gather_rows" in r[0])
    step = rows[first:]
    t0 = step[0][1]
    busy = sum(r[2] - r[1] for r in step)
    span = step[-1][2] - t0
    print("# last step of %s: %d launches, kernel time %.1f us, span %.1f us, idle between kernels %.1f us" % (
        path, len(step), busy / 1e3, span / 1e3, (span - busy) / 1e3))
    print("%4s %9s %8s %7s %7s  %s" % ("#", "start_us", "dur_us", "gap_us", "wgs", "kernel"))
    prev_end = t0
    for i, (name, s, e, gx, wx) in enumerate(step):
        print("%4d %9.1f %8.2f %7.2f %7d  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, gx // max(wx, 1),
                                                name.replace("void ", "").replace("vnr::", "")[:90]))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main(sys.argv[1])
