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


PHASES = [("hessenberg", ("hess_", "pack_yv", "conj_transpose_panel", "set_identity")), ("qr", ("qr_prepare", "qr_window", "apply_window", "qr_init")),
          ("schur_vectors", ("trevc", "colnorm"))]


def phases(path):
    """Wall-clock view of a kernel trace: per eigensolver phase (delimited by the first/last launch of its own kernels) the
    span, the time the GPU had at least one kernel running, and the idle remainder; everything else is 'other'."""
    con = sqlite3.connect(path)
    rows = sorted((start, end, short(name)) for name, start, end in con.cursor().execute("select name, start, end from kernels"))
    t0, t1 = rows[0][0], max(r[1] for r in rows)

    def union(iv):
        tot, cur_s, cur_e = 0, None, None
        for s_, e_ in sorted(iv):
            if cur_e is None or s_ > cur_e:
                if cur_e is not None:
                    tot += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        return tot + (cur_e - cur_s if cur_e is not None else 0)

    print(f"# {path}: trace span {(t1 - t0) / 1e9:.3f} s, GPU busy (union of kernels) {union([(a, b) for a, b, _ in rows]) / 1e9:.3f} s")
    # segment the trace: a phase instance = maximal run of launches until a kernel of ANOTHER phase shows up
    def phase_of(name):
        for ph, keys in PHASES:
            if any(k in name for k in keys):
                return ph
        return None
    segs, cur = [], None
    for a, b, nm in rows:
        ph = phase_of(nm)
        if ph is None:
            ph = cur[0] if cur and cur[0] in ("hessenberg", "schur_vectors") and "gemm" in nm else "other"
        if cur is None or ph != cur[0]:
            if cur:
                segs.append(cur)
            cur = [ph, a, b, [(a, b)]]
        else:
            cur[2] = max(cur[2], b)
            cur[3].append((a, b))
    segs.append(cur)
    agg = {}
    for ph, a, b, iv in segs:
        x = agg.setdefault(ph, [0, 0, 0])
        x[0] += 1
        x[1] += b - a
        x[2] += union(iv)
    print(f"{'phase':16s} {'segments':>9s} {'span_ms':>10s} {'busy_ms':>10s} {'idle_ms':>10s}")
    for ph, x in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{ph:16s} {x[0]:9d} {x[1] / 1e6:10.1f} {x[2] / 1e6:10.1f} {(x[1] - x[2]) / 1e6:10.1f}")


def gaps(path, top=25, min_us=50.0, from_ms=0.0):
    """Idle intervals of the GPU (no kernel running) longer than min_us: the host-side bubbles of the trace.
    from_ms: ignore everything before this offset (e.g. the warm-up step with its one-off allocations)."""
    con = sqlite3.connect(path)
    rows = sorted((start, end, short(name)) for name, start, end in con.cursor().execute("select name, start, end from kernels"))
    t00 = rows[0][0]
    rows = [r for r in rows if (r[0] - t00) / 1e6 >= from_ms]
    out, cur_end, cur_name = [], rows[0][1], rows[0][2]
    for a, b, nm in rows[1:]:
        if a > cur_end:
            if (a - cur_end) / 1e3 >= min_us:
                out.append((a - cur_end, cur_end - rows[0][0], cur_name, nm))
        if b > cur_end:
            cur_end, cur_name = b, nm
    tot = sum(g[0] for g in out)
    print(f"# {path}: {len(out)} idle gaps >= {min_us:.0f} us, total {tot / 1e6:.1f} ms of a {(rows[-1][1] - rows[0][0]) / 1e6:.1f} ms trace")
    print(f"{'gap_ms':>9s} {'at_ms':>10s}  after -> before")
    for g in sorted(out, key=lambda x: -x[0])[:top]:
        print(f"{g[0] / 1e6:9.3f} {g[1] / 1e6:10.1f}  {g[2]} -> {g[3]}")
    # histogram by (after, before) pair
    agg = {}
    for g in out:
        a = agg.setdefault((g[2], g[3]), [0, 0])
        a[0] += 1
        a[1] += g[0]
    print(f"{'count':>7s} {'total_ms':>9s}  after -> before")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{a[0]:7d} {a[1] / 1e6:9.1f}  {k[0]} -> {k[1]}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--gaps":
        gaps(sys.argv[1], from_ms=float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
    elif len(sys.argv) > 2 and sys.argv[2] == "--phases":
        phases(sys.argv[1])
    elif len(sys.argv) > 3 and sys.argv[2] == "--by-grid":
        by_grid(sys.argv[1], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
