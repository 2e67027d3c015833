#!/bin/bash
# full GPU suite + default bench line with the new roofline block + HW-queue experiment
R=$GRAFT_REPO_ROOT
cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r5m_gputests.txt; cat gpurun_out/r5m_gputests.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5m_bench.json 2> gpurun_out/r5m_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5m_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 1), "| dominant", r["kernel"], round(r["frac"], 3), "| Redheffer share", round(r["redheffer_share_of_step"], 3))
for k in r["kernels"]:
    print("  %-26s %-8s %8.1f ms/step  launches/step %7d  avg %9.1f us  achieved %12.4g %-28s frac %.3f" % (k["kernel"], k["bound"], k["est_total_ms_per_step"], k["launches"] // d["steps"], k["avg_us"], k["achieved"], k["unit"], k["frac"]))
for p in r["phases"]["phases"]:
    print("     %-62s %9.1f ms  %.3f" % (p["phase"], p["ms_per_step"], p["share_of_step"]))
for p in r["phases"]["inside_trx_eig"]:
    print("       %-60s %9.1f ms  %.3f" % (p["phase"], p["ms_per_step"], p["share_of_step"]))
PY
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* : "; env "$@" timeout 200 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line; }
run X=0
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=8 TRX_QR_GROUPS=8
run GPU_MAX_HW_QUEUES=8 TRX_QR_GROUPS=6
run TRX_QR_AED=48
B=16 run X=0
B=16 run GPU_MAX_HW_QUEUES=8 TRX_QR_GROUPS=4
