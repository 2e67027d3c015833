#!/bin/bash
# Counters of the QR update kernels (apply_links*) of one bench step, kernels serialised by rocprofv3 (--pmc): what are they waiting for?
#   usage: profiles/scripts/pmc_qr_updates.sh [batch] [extra env assignments ...]
cd /tmp && export TMPDIR=/tmp
B=${1:-128}; shift
RE='apply_links'
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmcq_$i
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$RE" -d /tmp/pmcq_$i -o g -- python $GRAFT_REPO_ROOT/bench.py --batch $B --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  i=$((i+1))
done
python - <<'PYEOF'
import sqlite3, glob, re
def short(name):
    m = re.search(r"([A-Za-z_0-9]+)<([^>(]*)", name)
    return f"{m.group(1)}<{m.group(2)}>" if (m and "trx" in name) else re.sub(r"\(.*", "", name)[:40]
agg = {}
for db in sorted(glob.glob("/tmp/pmcq_*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    try:
        rows = list(cur.execute("select name, duration, counter_name, counter_value from pmc_events"))
    except Exception as e:
        print("no pmc_events in", db, e); continue
    seen = set()
    for name, dur, cn, val in rows:
        a = agg.setdefault(short(name), {})
        a[cn] = a.get(cn, 0.0) + val
        a["dur_" + cn] = a.get("dur_" + cn, 0) + dur
        a["n_" + cn] = a.get("n_" + cn, 0) + 1
for k, a in agg.items():
    print("==", k)
    for cn in sorted(c for c in a if not c.startswith("dur_") and not c.startswith("n_")):
        print(f"   {cn:32s} sum {a[cn]:16.4g}   per launch {a[cn] / a['n_' + cn]:14.4g}   launches {a['n_' + cn]}   avg_us {a['dur_' + cn] / a['n_' + cn] / 1e3:9.1f}")
PYEOF
