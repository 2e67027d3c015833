#!/bin/bash
# L2 -> fabric traffic counters of the hot kernels of the bench step (one counter per pass, as MI355X_MICROARCH.md prescribes), per kernel.
#   usage: profiles/scripts/pmc_bench.sh [batch] [out.json]
# Counter collection is restricted to the kernels the roofline block quotes (--kernel-include-regex): with every one of the ~47 000 dispatches
# of a step instrumented a pass takes more than 7 minutes (rocprofv3 serialises and reads the counters back per dispatch; that is what cut
# off round 3's evidence call); the GEMM and Hessenberg-gemv kernels are ~3 000 dispatches.  The off-window update (apply_window, 17 000
# launches per step) is sampled by its own pass over a shorter command.
# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (guide, HBM section), so the table prints
# the raw value and the corrected traffic (2 F + W).
cd /tmp && export TMPDIR=/tmp
B=${1:-128}
OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/pmc_bench.json}
RE='gemm_big_kernel|gemm_mfma_kernel|hess_gemv_kernel'
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$RE" -d /tmp/pmcb_$c -o g -- python $GRAFT_REPO_ROOT/bench.py --batch $B --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
python - "$B" "$OUT" <<'PYEOF'
import sqlite3, glob, re, json, sys, os
batch, out = int(sys.argv[1]), sys.argv[2]
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from bench import csrc_sha16, csrc_file_hashes
def short(name):
    m = re.search(r"([A-Za-z_0-9]+)<([^>(]*)", name)
    return f"{m.group(1)}<{m.group(2)}>" if (m and "trx" in name) else re.sub(r"\(.*", "", name)[:40]
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob("/tmp/pmcb_%s/**/*.db" % c, recursive=True)
    if not dbs:
        print("no database for", c); sys.exit(1)
    cur = sqlite3.connect(dbs[0]).cursor()
    for name, dur, val in cur.execute("select name, duration, counter_value from pmc_events where counter_name = ?", (c,)):
        a = agg.setdefault(short(name), {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "dur_FETCH_SIZE": 0, "dur_WRITE_SIZE": 0, "n": 0})
        a[c] += val; a["dur_" + c] += dur
        if c == "FETCH_SIZE": a["n"] += 1
print(f"# bench.py --batch {batch} --steps 1 --warmup 0, one rocprofv3 --pmc pass per counter, counters on the GEMM and Hessenberg-gemv kernels only")
print(f"{'kernel':52s} {'calls':>7s} {'ms':>9s} {'FETCH_GB(raw)':>14s} {'WRITE_GB':>10s} {'(F+W)/t GB/s':>13s} {'(2F+W)/t GB/s':>14s} {'MB/launch(2F+W)':>16s}")
js = {"batch": batch, "csrc_sha16": csrc_sha16(), "src_sha16": csrc_file_hashes(), "steps_traced": 1,
      "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-include-regex 'gemm_big_kernel|gemm_mfma_kernel|hess_gemv_kernel' -- python bench.py --batch %d --steps 1 --warmup 0 --no-cpu-baseline" % batch,
      "note": "KB counters summed over all launches of a kernel; corrected = 2*FETCH + WRITE (gfx950 FETCH_SIZE under-reports coalesced reads 2x); ms = the kernel running ALONE (serialised under --pmc)", "kernels": {}}
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur_FETCH_SIZE"])[:24]:
    t = a["dur_FETCH_SIZE"] * 1e-9
    f, w = a["FETCH_SIZE"] * 1024 / 1e9, a["WRITE_SIZE"] * 1024 / 1e9
    print(f"{k:52s} {a['n']:7d} {t*1e3:9.1f} {f:14.1f} {w:10.1f} {(f+w)/max(t,1e-9):13.0f} {(2*f+w)/max(t,1e-9):14.0f} {(2*f+w)*1e3/max(a['n'],1):16.1f}")
    js["kernels"][k] = {"launches": a["n"], "ms_total": t * 1e3, "total_ms": t * 1e3, "fetch_GB_raw": f, "write_GB": w,
                        "bytes_per_launch_raw": (f + w) * 1e9 / max(a["n"], 1), "bytes_per_launch_corrected": (2 * f + w) * 1e9 / max(a["n"], 1)}
json.dump(js, open(out, "w"), indent=1)
PYEOF
