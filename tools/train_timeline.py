#!/usr/bin/env python3
"""One training step out of a rocprofv3 (rocpd sqlite) kernel trace of tools/bench_train.py: which stream is busy when.
A step ends with adam_kernel; the step before the last one is analysed (steady state).  Prints, per queue / stream: kernel time,
the span, the idle time in front of kernels, and per kernel name the time on the busiest (main) stream -- what the step's wall
time is made of.

usage: python tools/train_timeline.py trace_results.db [index of the step's adam launch, default -2]"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    sel = "name, start, end, %s" % (qcol or "0")
    rows = c.execute("select %s from kernels order by start" % sel).fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    if len(adam) < 3:
        print("need >= 3 steps in the trace"); return
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -2       # index of the adam launch that ENDS the analysed step
    a0, a1 = adam[which - 1], adam[which]
    step = rows[a0 + 1:a1 + 1]
    t0, t1 = rows[a0][2], step[-1][2]
    print("# columns: %s; stream key: %s" % (",".join(cols), qcol))
    print("# step: %d launches, span %.2f ms" % (len(step), (t1 - t0) / 1e6))
    by_q = defaultdict(list)
    for r in step:
        by_q[r[3]].append(r)
    main_q = max(by_q, key=lambda q: sum(r[2] - r[1] for r in by_q[q]))
    for q, rs in sorted(by_q.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        busy = sum(r[2] - r[1] for r in rs)
        gaps, prev = 0, None
        for r in rs:
            if prev is not None and r[1] > prev:
                gaps += r[1] - prev
            prev = max(prev or 0, r[2])
        print("stream %s%s: %4d launches, kernel time %.2f ms, first start +%.2f ms, last end +%.2f ms, idle between its kernels %.2f ms"
              % (q, " (main)" if q == main_q else "", len(rs), busy / 1e6, (rs[0][1] - t0) / 1e6, (rs[-1][2] - t0) / 1e6, gaps / 1e6))
    # union of busy intervals over all streams
    ev = sorted((r[1], r[2]) for r in step)
    u, cs, ce = 0, None, None
    for s, e in ev:
        if ce is None or s > ce:
            if ce is not None:
                u += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    u += ce - cs
    print("# some kernel running: %.2f ms of the %.2f ms span (GPU fully idle %.2f ms)" % (u / 1e6, (t1 - t0) / 1e6, (t1 - t0 - u) / 1e6))
    # main stream: gaps by size, and time by kernel name
    rs = by_q[main_q]
    gl = []
    prev = t0
    prev_name = "(step start)"
    for i, r in enumerate(rs):
        if r[1] > prev:
            gl.append((r[1] - prev, r[0] + "   <- after #%d %s at +%.2f ms" % (i, prev_name.replace("void ", "").replace("vnr::", "").split("(")[0], (r[1] - t0) / 1e6)))
        prev = max(prev, r[2]); prev_name = r[0]
    # the longest gap of the main stream: what the other streams ran meanwhile
    if gl:
        prev = t0
        big = (0, 0, 0)
        for r in rs:
            if r[1] - prev > big[0]:
                big = (r[1] - prev, prev, r[1])
            prev = max(prev, r[2])
        win = [r for r in step if r[2] > big[1] - 100000 and r[1] < big[2] + 100000]
        print("# the main stream's longest gap: +%.2f .. +%.2f ms; launches of all streams around it (first 60):" % ((big[1] - t0) / 1e6, (big[2] - t0) / 1e6))
        for r in win[:60]:
            print("#   stream %s  +%8.3f .. +%8.3f ms  %s" % (r[3], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, r[0].replace("void ", "").replace("vnr::", "").split("(")[0][:60]))
    gl.sort(reverse=True)
    tot_gap = sum(g for g, _ in gl)
    print("# main stream: %d gaps, %.2f ms in total; gaps > 20 us: %d (%.2f ms); the ten longest (us, kernel that followed):" % (
        len(gl), tot_gap / 1e6, sum(1 for g, _ in gl if g > 20000), sum(g for g, _ in gl if g > 20000) / 1e6))
    for g, n in gl[:10]:
        print("   %8.1f  %s" % (g / 1e3, n.replace("void ", "").replace("vnr::", "")[:170]))
    agg = defaultdict(lambda: [0, 0])
    for r in rs:
        k = r[0].replace("void ", "").replace("vnr::", "").split("(")[0]
        agg[k][0] += 1; agg[k][1] += r[2] - r[1]
    print("# main stream by kernel (calls, ms):")
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("   %-70s %5d %8.3f" % (k[:70], n, d / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
