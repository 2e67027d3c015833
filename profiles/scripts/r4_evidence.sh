#!/bin/bash
# Round 4 evidence call (run on the sources the round ends with):   bash profiles/scripts/r4_evidence.sh
#   1. kernel trace of the bench command      -> profiles/r04_kernel_profile.json + gpurun_out/r04_bench_final_*.txt
#   2. counter passes (FETCH_SIZE, WRITE_SIZE) -> profiles/r04_pmc_bench.json / gpurun_out/r04_pmc_bench.txt
#   3. the default bench line, which quotes 1 and 2 (frac_rocprof, frac_alone, traffic) -> gpurun_out/r04_bench_final.json
#   4. small batches (strong-scaling table)
R=$GRAFT_REPO_ROOT
cd $R
bash profiles/scripts/trace_bench.sh r04_bench_final
cp gpurun_out/r04_bench_final_kernel_profile.json profiles/r04_kernel_profile.json
bash profiles/scripts/pmc_bench.sh 128 $R/gpurun_out/r04_pmc_bench.json > gpurun_out/r04_pmc_bench.txt 2>&1
cp gpurun_out/r04_pmc_bench.json profiles/r04_pmc_bench.json
head -8 gpurun_out/r04_pmc_bench.txt
timeout 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
python - <<'PYEOF'
import json
d = json.loads(open("gpurun_out/r04_bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d.get("parity_sample", {}).get("max_rel_err_vs_c128_oracle"), "cpu", d.get("cpu_baseline", {}).get("value"))
print("dominant", r["kernel"], "frac", r["frac"], "frac_rocprof", r.get("frac_rocprof"), "frac_alone", r.get("frac_alone"), "traffic", r.get("traffic"), r.get("bound"), r.get("bound_note"))
for k in r["kernels"]:
    print("  %-28s %7.1f ms/step  frac %.3f  rocprof %s" % (k["kernel"], k["est_total_ms_per_step"], k["frac"], k.get("frac_rocprof")))
PYEOF
export TRX_BENCH_NOPROF=1
for b in 16 32 64; do
  timeout 200 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null > gpurun_out/r04_bench_b$b.json
  python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_b$b.json').read().strip().splitlines()[-1]); print('batch $b', round(d['value'],2), 'layer-solves/s', round(d['ms_per_step'],1), 'ms')"
done
