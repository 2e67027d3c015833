#!/bin/bash
# Round 6 evidence call (run on the sources the round ends with):   bash profiles/scripts/r6_evidence.sh
#   1. kernel trace of the bench command      -> profiles/r06_kernel_profile.json + gpurun_out/r06_bench_final_*.txt
#   2. counter passes (FETCH_SIZE, WRITE_SIZE) of the GEMM / Hessenberg kernels -> profiles/r06_pmc_bench.json ; counters of the QR update kernels
#   3. the default bench line, which quotes 1 and 2 (frac_rocprof, frac_alone, traffic) -> gpurun_out/r06_bench_final.json
R=$GRAFT_REPO_ROOT
cd $R
bash profiles/scripts/trace_bench.sh r06_bench_final
cp gpurun_out/r06_bench_final_kernel_profile.json profiles/r06_kernel_profile.json
bash profiles/scripts/pmc_bench.sh 128 $R/gpurun_out/r06_pmc_bench.json > gpurun_out/r06_pmc_bench.txt 2>&1
cp gpurun_out/r06_pmc_bench.json profiles/r06_pmc_bench.json
head -8 gpurun_out/r06_pmc_bench.txt
bash profiles/scripts/pmc_qr_updates.sh 128 > gpurun_out/r06_pmc_qr_updates.txt 2>&1
timeout 900 python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
python - <<'PYEOF'
import json
d = json.loads(open("gpurun_out/r06_bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
dk = r["dominant_kernel"]
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d.get("parity_sample", {}).get("max_rel_err_vs_c128_oracle"), "cpu", d.get("cpu_baseline", {}).get("value"))
print("headline (whole layer-solve, SURVEY 8(d)): frac", r["frac"], "achieved", r["achieved"], r["unit"], "| at the fp64 peak", r["layer_solve"]["frac_at_fp64_peak"])
print("dominant kernel", dk["kernel"], "frac", dk["frac"], "frac_rocprof", dk.get("frac_rocprof"), "frac_alone", dk.get("frac_alone"), "traffic", dk.get("traffic"), dk.get("bound"), dk.get("bound_note"))
for k in r["kernels"]:
    print("  %-28s %-8s %7.1f ms/step  frac %.3f  rocprof %s" % (k["kernel"], k["bound"], k["est_total_ms_per_step"], k["frac"], k.get("frac_rocprof")))
for p in r["phases"]["phases"] + r["phases"]["inside_trx_eig"]:
    print("     %-62s %9.1f ms  %.3f" % (p["phase"], p["ms_per_step"], p["share_of_step"]))
PYEOF
