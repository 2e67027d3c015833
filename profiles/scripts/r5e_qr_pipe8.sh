run() { # name batch env...
  name=$1; b=$2; shift 2
  env "$@" timeout 200 python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5e_$name.json 2> gpurun_out/r5e_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5e_$name.json").read().strip().splitlines()[-1])
    ks={k["kernel"]:(k["launches"],round(k["avg_us"],1)) for k in d["roofline"]["kernels"]}
    print("$name", "$*", round(d["value"],2), round(d["ms_per_step"],1), {k:v for k,v in ks.items() if "qr" in k or "apply" in k})
except Exception as e: print("$name ERR", e)
PY
}
(timeout 300 python -m pytest tests/test_eig.py -m gpu -x -q -k "super_steps or mixed or random" 2>&1 | tail -3)
run b128_p0 128 TRX_SLAB_PIPE=0
run b128_p2 128 TRX_SLAB_PIPE=2
run b128_p0_s8 128 TRX_SLAB_PIPE=0 TRX_QR_SUPER=8
run b16_p0 16 TRX_SLAB_PIPE=0
run b16_p2 16 TRX_SLAB_PIPE=2
bash profiles/scripts/trace_bench.sh r5e_b128 --batch 128 2>&1 | tail -1
head -12 gpurun_out/r5e_b128_kernel_stats.txt; cat gpurun_out/r5e_b128_phases.txt; head -6 gpurun_out/r5e_b128_lanes.txt | cut -c1-330
