#!/bin/bash
# Round 6, first GPU call: CU-mask bit order probe, the GPU test suite on the new refinement / Hessenberg sub-batches, A/B of the
# sub-batch split, cluster statistics of configs 3 and 4, cycle counters of the QR chain kernels.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call1.txt
: > $O
echo "== cumask probe" >> $O
timeout 120 tests/micro/_build/cumask_probe >> $O 2>&1
echo "== gpu tests" >> $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph={p['phase'].split(':')[-1].strip(): round(p['ms_per_step']) for p in d['roofline']['phases']['inside_trx_eig']} if d.get('roofline') and d['roofline'].get('phases') else {}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), d.get('parity_sample',{}).get('max_rel_err_vs_c128_oracle'), ph)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call1.err | line >> $O; }
echo "== bench A/B (phases inside trx_eig in ms per step)" >> $O
run X=default
run TRX_HESS_SPLIT=1
run TRX_HESS_SPLIT=3
run TRX_HESS_SPLIT=4
B=16 run X=default
B=16 run TRX_HESS_SPLIT=1
B=64 run X=default
echo "== default line with parity sample (cpu baseline, 1 point)" >> $O
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-points 1 > gpurun_out/r6_call1_bench_default.json 2>>gpurun_out/r6_call1.err
python - >> $O <<'PYEOF'
import json
d = json.loads(open("gpurun_out/r6_call1_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d.get("parity_sample"), "cpu", d.get("cpu_baseline", {}).get("value"))
r = d["roofline"]
for k in r["kernels"]:
    print("  %-28s %-8s %7.1f ms/step  frac %.3f" % (k["kernel"], k["bound"], k["est_total_ms_per_step"], k["frac"]))
for p in r["phases"]["phases"] + r["phases"]["inside_trx_eig"]:
    print("     %-62s %9.1f ms  %.3f" % (p["phase"], p["ms_per_step"], p["share_of_step"]))
PYEOF
echo "== cluster statistics: config 4 (first 512 points), config 3 (one step)" >> $O
TRX_EIG_DEBUG=1 timeout 600 python bench.py --config 4 --points 512 --steps 1 --warmup 0 --no-cpu-baseline --eig-route mixed 2> gpurun_out/r6_call1_cfg4_stats.err | line >> $O
grep "eig_refine" gpurun_out/r6_call1_cfg4_stats.err | cut -c1-1500 >> $O
TRX_EIG_DEBUG=1 timeout 900 python bench.py --config 3 --steps 1 --warmup 0 --no-cpu-baseline --eig-route mixed 2> gpurun_out/r6_call1_cfg3_stats.err | line >> $O
grep "eig_refine" gpurun_out/r6_call1_cfg3_stats.err | cut -c1-1500 >> $O
echo "== config 4 / 3 / 5 on auto" >> $O
FLAGS="--config 4 --points 512" run X=auto
FLAGS="--config 5" B=1 run X=auto
echo "== QR cycle counters (batch 16)" >> $O
TRX_QR_DEBUG=1 timeout 300 python bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "libtrx qr" | tail -3 >> $O
tail -50 gpurun_out/r6_call1.err > gpurun_out/r6_call1_errtail.txt
cat $O
