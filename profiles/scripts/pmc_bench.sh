#!/bin/bash
# HBM traffic counters of the whole hot path (one counter per pass), aggregated per kernel.
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcb_$c -o g -- python $GRAFT_REPO_ROOT/bench.py --batch ${1:-32} --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
python - <<'PYEOF'
import sqlite3, glob, re
def short(name):
    m = re.search(r"([A-Za-z_0-9]+)<([^>(]*)", name)
    return f"{m.group(1)}<{m.group(2)}>" if (m and "trx" in name) else re.sub(r"\(.*", "", name)[:40]
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob("/tmp/pmcb_%s/**/*.db" % c, recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    for name, dur, val in cur.execute("select name, duration, counter_value from pmc_events where counter_name = ?", (c,)):
        a = agg.setdefault(short(name), {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "dur_FETCH_SIZE": 0, "dur_WRITE_SIZE": 0, "n": 0})
        a[c] += val; a["dur_" + c] += dur
        if c == "FETCH_SIZE": a["n"] += 1
print(f"{'kernel':44s} {'calls':>7s} {'ms':>9s} {'FETCH_GB(raw)':>14s} {'WRITE_GB':>10s} {'(F+W)/t GB/s':>13s} {'(2F+W)/t GB/s':>14s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur_FETCH_SIZE"])[:12]:
    t = a["dur_FETCH_SIZE"] * 1e-9
    f, w = a["FETCH_SIZE"] * 1024 / 1e9, a["WRITE_SIZE"] * 1024 / 1e9
    print(f"{k:44s} {a['n']:7d} {t*1e3:9.1f} {f:14.1f} {w:10.1f} {(f+w)/t:13.0f} {(2*f+w)/t:14.0f}")
PYEOF
