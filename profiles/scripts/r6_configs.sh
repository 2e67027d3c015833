#!/bin/bash
# Round 6: the other GPU configurations through bench.py (each line carries `roofline` with kernels, phases and the Redheffer share)
#   usage: bash profiles/scripts/r5_configs.sh        -> gpurun_out/r06_bench_config{3,4,5}.json, r06_bench_b{16,32,64}.json
R=$GRAFT_REPO_ROOT
cd $R
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"], 3), d["unit"], round(d["ms_per_step"], 1), "ms/step | dominant", (r.get("dominant_kernel") or {}).get("kernel"), "frac", round((r.get("dominant_kernel") or {}).get("frac", 0), 3),
          "| layer-solve frac (fp64 / fp32 peak)", round(r.get("layer_solve", {}).get("frac_at_fp64_peak", 0), 3), round(r.get("layer_solve", {}).get("frac_at_fp32_peak", 0), 3),
          "| Redheffer share", round(r.get("redheffer_share_of_step", 0), 3), "| eig route:", d["config"].get("eig_route", "")[:90], "| hbm", d.get("hbm", {}).get("peak_reserved_GB"))
    for p in (r.get("phases") or {}).get("phases", []):
        print("     %-60s %9.1f ms  %.3f" % (p["phase"], p["ms_per_step"], p["share_of_step"]))
    for p in (r.get("phases") or {}).get("inside_trx_eig", []):
        print("       %-58s %9.1f ms  %.3f" % (p["phase"], p["ms_per_step"], p["share_of_step"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
timeout 900 python bench.py --config 4 --points 512 --steps 2 --warmup 1 > gpurun_out/r06_bench_config4.json 2> gpurun_out/r06_bench_config4.err; show gpurun_out/r06_bench_config4.json
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/r06_bench_config5.json 2> gpurun_out/r06_bench_config5.err; show gpurun_out/r06_bench_config5.json
timeout 1500 python bench.py --config 3 --steps 1 --warmup 1 > gpurun_out/r06_bench_config3.json 2> gpurun_out/r06_bench_config3.err; show gpurun_out/r06_bench_config3.json
for b in 16 32 64; do
  timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg > gpurun_out/r06_bench_b$b.json 2>/dev/null; show gpurun_out/r06_bench_b$b.json | head -1
done
