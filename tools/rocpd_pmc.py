#!/usr/bin/env python3
"""Per-kernel PMC table from a rocprofv3 --pmc run (rocpd sqlite).  usage: rocpd_pmc.py file.db [name-filter]"""
import collections
import sqlite3
import sys


def main(path, filt=""):
    c = sqlite3.connect(path)
    acc = collections.OrderedDict()
    for name, gx, disp, cn, val, dur in c.execute(
            "select kernel_name, grid_size, dispatch_id, counter_name, sum(value), max(duration) from counters_collection "
            "group by dispatch_id, counter_name order by dispatch_id"):
        if filt and filt not in name:
            continue
        key = (name.replace("void ", "")[:60], gx)
        d = acc.setdefault(key, collections.defaultdict(float))
        d[cn] += val
        d["_n_" + cn] += 1
        d["_dur"] = dur
    names = sorted({k for d in acc.values() for k in d if not k.startswith("_")})
    print("%-60s %9s %8s " % ("kernel", "grid", "dur_us") + " ".join("%14s" % n[-14:] for n in names))
    for (k, gx), d in acc.items():
        print("%-60s %9d %8.1f " % (k, gx, d["_dur"] / 1e3) + " ".join("%14.4g" % (d[n] / max(1, d["_n_" + n])) for n in names))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
