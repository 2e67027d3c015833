#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel table (calls, total, avg, share)."""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"([A-Za-z_0-9]+)<([^>(]*)", name)
    if m and "trx" in name:
        return f"{m.group(1)}<{m.group(2)}>"
    return re.sub(r"\(.*", "", name)[:60]


def main(path, top=30):
    con = sqlite3.connect(path)
    cur = con.cursor()
    agg = {}
    for name, start, end in cur.execute("select name, start, end from kernels"):
        k = short(name)
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += end - start
    tot = sum(a[1] for a in agg.values())
    print(f"# {path}: total kernel time {tot / 1e9:.3f} s over {sum(a[0] for a in agg.values())} dispatches")
    print(f"{'kernel':58s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:58s} {a[0]:8d} {a[1] / 1e6:10.1f} {a[1] / a[0] / 1e3:10.1f} {100 * a[1] / tot:6.1f}%")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
