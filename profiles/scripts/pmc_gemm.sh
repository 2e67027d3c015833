cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o g -- python $GRAFT_REPO_ROOT/tests/gpu_gemm_pmc.py 2>&1 | grep "TFLOP"; done
ls -R /tmp/pmc_FETCH_SIZE | head -5
python - <<'PYEOF'
import sqlite3, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob("/tmp/pmc_%s/**/*.db" % c, recursive=True)[0]
    con = sqlite3.connect(db); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cand = [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]
    print(c, cand[:12])
    for t in cand:
        if not t.startswith("rocpd_"):
            cols = [d[1] for d in cur.execute("pragma table_info('%s')" % t)]
            print("  view", t, cols)
            try:
                for row in list(cur.execute("select * from '%s' limit 8" % t)):
                    print("     ", [str(x)[:50] for x in row])
            except Exception as e:
                print("     err", e)
PYEOF
