#!/bin/bash
# Copy the summaries of the final call (gpurun_out/, scratch) into profiles/ under the names profiles/README.md lists.
cd $(git rev-parse --show-toplevel)
G=gpurun_out; P=profiles
for f in r06_bench_final.json r06_bench_final_concurrency.txt r06_bench_final_excerpt.txt r06_bench_final_gaps.txt r06_bench_final_gemm_by_shape.txt \
         r06_bench_final_kernel_stats.txt r06_bench_final_lanes.txt r06_bench_final_phases.txt r06_bench_config3.json r06_bench_config4.json r06_bench_config5.json \
         r06_bench_b16.json r06_bench_b32.json r06_bench_b64.json r06_bench_native.json r06_bench_fp64route.json r06_final.txt r06_gputests_tail.txt \
         r06_pmc_bench.json r06_pmc_bench.txt r06_pmc_qr_updates.txt; do
  [ -s $G/$f ] && cp $G/$f $P/$f
done
[ -s $G/r06_bench_final_kernel_profile.json ] && cp $G/r06_bench_final_kernel_profile.json $P/r06_kernel_profile.json
[ -s $G/r06_bench_final_bench.json ] && cp $G/r06_bench_final_bench.json $P/r06_bench_under_trace.json
[ -s $G/r06_bench_final_stats.csv ] && cp $G/r06_bench_final_stats.csv $P/r06_bench_final_rocprofv3_stats.csv
python - <<'PY'
import json
d=json.loads(open("profiles/r06_bench_final.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("default line:", round(d["value"],2), d["unit"], round(d["ms_per_step"],1), "ms; headline frac", round(r["frac"],4), "| dominant", r["dominant_kernel"]["kernel"], round(r["dominant_kernel"]["frac"],3), "rocprof", r["dominant_kernel"].get("frac_rocprof"), "| cpu", d["cpu_baseline"]["value"], "| parity", d.get("parity_sample",{}).get("max_rel_err") if isinstance(d.get("parity_sample"),dict) else d.get("parity_sample"))
PY
