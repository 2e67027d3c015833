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


PHASES = [("hessenberg", ("hess_", "pack_yv", "conj_transpose_panel", "set_identity")), ("qr", ("qr_prepare", "qr_window", "apply_window", "apply_links", "qr_init")),
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


def concurrency(path, pattern="qr_|apply_window|apply_links"):
    """Time-weighted histogram of the number of kernels running at once while at least one kernel matching `pattern` runs, the
    same per stream/queue, and the in-stream gap between consecutive kernels of one stream (launch latency the GPU sees)."""
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.cursor().execute("pragma table_info(kernels)")]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    q = f"select name, start, end, {qcol if qcol else '0'} from kernels"
    rows = sorted((a, b, short(nm), sid) for nm, a, b, sid in con.cursor().execute(q))
    pat = re.compile(pattern)
    sel = [r for r in rows if pat.search(r[2])]
    t_lo, t_hi = sel[0][0], max(r[1] for r in sel)
    ev = []
    for a, b, nm, sid in rows:
        if b < t_lo or a > t_hi:
            continue
        ev.append((a, 1, nm))
        ev.append((b, -1, nm))
    ev.sort()
    hist, cur, last = {}, 0, ev[0][0]
    by_kind = {}
    active = {}
    for tme, d, nm in ev:
        hist[cur] = hist.get(cur, 0) + (tme - last)
        key = tuple(sorted(k for k, v in active.items() if v > 0))
        by_kind[key] = by_kind.get(key, 0) + (tme - last)
        last = tme
        cur += d
        active[nm.split("<")[0]] = active.get(nm.split("<")[0], 0) + d
    tot = sum(hist.values())
    print(f"# {path}: QR-phase window {(t_hi - t_lo) / 1e6:.1f} ms; stream column: {qcol}")
    print("concurrent kernels : share of time")
    for k in sorted(hist):
        print(f"   {k:2d} : {100 * hist[k] / tot:5.1f}%")
    print("kernel kinds running together (top 12):")
    for k, v in sorted(by_kind.items(), key=lambda kv: -kv[1])[:12]:
        print(f"   {100 * v / tot:5.1f}%  {' + '.join(k) if k else '(idle)'}")
    if qcol:
        per = {}
        for a, b, nm, sid in rows:
            if a < t_lo or a > t_hi:
                continue
            per.setdefault(sid, []).append((a, b, nm))
        print("per stream: kernels, busy ms, in-stream gap (median / mean us)")
        for sid, lst in sorted(per.items()):
            gaps_ = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
            gaps_ = [g for g in gaps_ if g < 5e6]
            busy = sum(b - a for a, b, _ in lst)
            if gaps_:
                gs = sorted(gaps_)
                print(f"   stream {sid}: {len(lst)} kernels, busy {busy / 1e6:.1f} ms, gap median {gs[len(gs) // 2] / 1e3:.1f} us, mean {sum(gs) / len(gs) / 1e3:.1f} us")


def excerpt(path, at_frac=0.55, span_us=1500.0):
    """Raw timeline excerpt: every kernel that runs within span_us after the point at_frac of the trace (stream, start, duration)."""
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.cursor().execute("pragma table_info(kernels)")]
    qcol = "stream_id" if "stream_id" in cols else "0"
    rows = sorted((a, b, short(nm), sid, gx // max(wx, 1), gy // max(wy, 1)) for nm, a, b, sid, gx, gy, wx, wy in
                  con.cursor().execute(f"select name, start, end, {qcol}, grid_x, grid_y, workgroup_x, workgroup_y from kernels"))
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # start the excerpt at an apply_window kernel near at_frac
    lo = t0 + at_frac * (t1 - t0)
    cand = [r for r in rows if r[0] >= lo and ("apply_window" in r[2] or "apply_links" in r[2])]
    lo = cand[0][0] if cand else lo
    print(f"# timeline excerpt, {span_us:.0f} us from t = {(lo - t0) / 1e6:.1f} ms; columns: stream  start_us  dur_us  kernel  grid")
    for a, b, nm, sid, gx, gy in rows:
        if b >= lo and a <= lo + span_us * 1e3:
            print(f"  s{sid}  {(a - lo) / 1e3:9.1f}  {(b - a) / 1e3:8.1f}  {nm.split('<')[0]:24s} ({gx},{gy})")


def lanes(path, at_frac=0.55, span_ms=30.0, bucket_us=100.0):
    """One text row per stream over span_ms from the point at_frac of the trace, one character per bucket of bucket_us: the kernel
    kind that was busy longest in the bucket (P qr_prepare, W qr_window, A apply_window, G gemm, H hess_*, L lu_* / trsm, o other,
    '.' idle).  Shows at a glance which iteration groups of the QR phase run and which wait."""
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.cursor().execute("pragma table_info(kernels)")]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = sorted((a, b, short(nm), sid) for nm, a, b, sid in con.cursor().execute(f"select name, start, end, {qcol} from kernels"))
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + at_frac * (t1 - t0)
    cand = [r for r in rows if r[0] >= lo and "qr_window" in r[2]]        # start inside a QR phase (a little after its beginning)
    lo = cand[min(len(cand) - 1, 200)][0] if cand else lo
    hi = lo + span_ms * 1e6
    nb = int(span_ms * 1e3 / bucket_us)
    kinds = (("qr_prepare", "P"), ("qr_window", "W"), ("apply_window", "A"), ("apply_links_f32_kernel<1", "R"), ("apply_links_kernel<float, 1", "R"), ("apply_links_kernel<double, 1", "R"),
             ("apply_links", "A"), ("gemm", "G"), ("hess_", "H"), ("lu_", "L"), ("trsm", "L"))
    per = {}
    for a, b, nm, sid in rows:
        if b < lo or a > hi:
            continue
        ch = next((c for key, c in kinds if key in nm), "o")
        acc = per.setdefault(sid, [dict() for _ in range(nb)])
        i0, i1 = max(0, int((a - lo) / (bucket_us * 1e3))), min(nb - 1, int((b - lo) / (bucket_us * 1e3)))
        for i in range(i0, i1 + 1):
            b_lo, b_hi = lo + i * bucket_us * 1e3, lo + (i + 1) * bucket_us * 1e3
            ov = min(b, b_hi) - max(a, b_lo)
            if ov > 0:
                acc[i][ch] = acc[i].get(ch, 0) + ov
    print(f"# stream lanes, {span_ms:.0f} ms from t = {(lo - t0) / 1e6:.1f} ms, one character per {bucket_us:.0f} us (stream column: {qcol})")
    for sid in sorted(per):
        row = "".join((max(d.items(), key=lambda kv: kv[1])[0] if d else ".") for d in per[sid])
        busy = sum(sum(d.values()) for d in per[sid]) / (span_ms * 1e6)
        print(f"  s{sid} {100 * busy:5.1f}% |{row}|")


def to_json(path, out, batch, command):
    """Per-kernel averages of the trace as JSON, stamped with the hash of the kernel sources (bench.py reads it back for the
    `frac_rocprof` figure of its roofline block and refuses it on other sources)."""
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import re
    from bench import csrc_sha16, csrc_file_hashes
    m_steps, m_warm = re.search(r"--steps (\d+)", command), re.search(r"--warmup (\d+)", command)
    steps_traced = (int(m_steps.group(1)) if m_steps else 1) + (int(m_warm.group(1)) if m_warm else 0)
    agg = {}
    for name, start, end in sqlite3.connect(path).cursor().execute("select name, start, end from kernels"):
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += end - start
    js = {"csrc_sha16": csrc_sha16(), "src_sha16": csrc_file_hashes(), "batch": int(batch), "command": command, "steps_traced": steps_traced,
          "kernels": {k: {"launches": a[0], "avg_us": a[1] / a[0] / 1e3, "total_ms": a[1] / 1e6} for k, a in agg.items()}}
    json.dump(js, open(out, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[2] == "--json":
        to_json(sys.argv[1], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
    elif len(sys.argv) > 2 and sys.argv[2] == "--excerpt":
        excerpt(sys.argv[1], float(sys.argv[3]) if len(sys.argv) > 3 else 0.55)
    elif len(sys.argv) > 2 and sys.argv[2] == "--lanes":
        lanes(sys.argv[1], float(sys.argv[3]) if len(sys.argv) > 3 else 0.55, float(sys.argv[4]) if len(sys.argv) > 4 else 30.0)
    elif len(sys.argv) > 2 and sys.argv[2] == "--concurrency":
        concurrency(sys.argv[1])
    elif len(sys.argv) > 2 and sys.argv[2] == "--gaps-frac":
        con = sqlite3.connect(sys.argv[1])
        t0, t1 = con.cursor().execute("select min(start), max(end) from kernels").fetchone()
        gaps(sys.argv[1], from_ms=float(sys.argv[3]) * (t1 - t0) / 1e6, min_us=20.0)
    elif len(sys.argv) > 2 and sys.argv[2] == "--gaps":
        gaps(sys.argv[1], from_ms=float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
    elif len(sys.argv) > 2 and sys.argv[2] == "--phases":
        phases(sys.argv[1])
    elif len(sys.argv) > 3 and sys.argv[2] == "--by-grid":
        by_grid(sys.argv[1], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
