#!/bin/bash
# Counter passes of the hot GEMM shape alone (1922^3 x 128, complex128): L2-miss traffic, L2 hit rate, wave-cycle breakdown, MFMA busy.
#   usage: bash profiles/scripts/pmc_gemm_hot.sh [tag]      -> stdout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-g}
pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pg_${TAG}_$name
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pg_${TAG}_$name -o g -- python $R/tests/gpu_gemm_pmc.py 1922 1922 1922 128 2>&1 | grep "TFLOP\|rror" | head -3
  python - /tmp/pg_${TAG}_$name <<'PYEOF'
import sqlite3, glob, sys
dbs = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
if not dbs: print("  no database"); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
agg = {}
for name, dur, cn, val in cur.execute("select name, duration, counter_name, counter_value from pmc_events"):
    if "gemm" not in name: continue
    a = agg.setdefault(cn, [0.0, 0, 0]); a[0] += val; a[1] += dur; a[2] += 1
for cn, (v, d, n) in sorted(agg.items()):
    print(f"  {cn:32s} per launch {v/n:16.1f}   launches {n}   avg duration {d/n/1e3:10.1f} us")
PYEOF
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64
pass sq2 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
