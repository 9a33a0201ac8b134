#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the text summary committed under profiles/.

usage: python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_kernel_stats.txt
Per kernel: calls, total/avg/min/max duration (us), share of GPU kernel time, plus grid/LDS/VGPR of
the most frequent launch shape.  Durations are rocprofv3's (end - start) in ns.
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
        "max(workgroup_x*workgroup_y*workgroup_z) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# total GPU kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    print("%-86s %7s %11s %9s %9s %9s %6s %7s %5s %5s %5s %4s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "lds", "vgpr", "agpr", "sgpr", "wg"))
    for r in rows:
        name = r[0].replace("void ", "")
        print("%-86s %7d %11.1f %9.2f %9.2f %9.2f %6.2f %7d %5d %5d %5d %4d" % (
            name[:86], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0))
    print()
    print("# per launch shape (kernel, grid, calls, avg_us)")
    for r in c.execute(
            "select name, grid_x, grid_y, grid_z, count(*), avg(duration) from kernels "
            "group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 60"):
        print("%-86s grid=(%d,%d,%d) calls=%d avg_us=%.2f" % (r[0].replace("void ", "")[:86], r[1], r[2], r[3], r[4], r[5] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
