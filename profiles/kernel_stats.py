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


def by_grid(path, pattern, top=40):
    """Per launch geometry (grid in workgroups) of the kernels whose name contains `pattern`: the shape histogram of a GEMM."""
    con = sqlite3.connect(path)
    agg = {}
    q = "select name, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, start, end from kernels"
    for name, gx, gy, gz, wx, wy, wz, start, end in con.cursor().execute(q):
        if pattern not in name:
            continue
        k = (short(name), gx // max(wx, 1), gy // max(wy, 1), gz // max(wz, 1))
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += end - start
    tot = sum(a[1] for a in agg.values())
    print(f"# {path}: kernels matching '{pattern}': {tot / 1e9:.3f} s over {sum(a[0] for a in agg.values())} dispatches")
    print(f"{'kernel':40s} {'grid (workgroups)':>20s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k[0]:40s} {str(k[1:]):>20s} {a[0]:7d} {a[1] / 1e6:10.1f} {a[1] / a[0] / 1e3:10.1f} {100 * a[1] / tot:6.1f}%")


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--by-grid":
        by_grid(sys.argv[1], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
