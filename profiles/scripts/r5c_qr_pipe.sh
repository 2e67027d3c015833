set -x
(timeout 600 python -m pytest tests/test_eig.py -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r5c_eigtests.txt
run() { # name batch env...
  name=$1; b=$2; shift 2
  env "$@" timeout 200 python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5c_$name.json 2> gpurun_out/r5c_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5c_$name.json").read().strip().splitlines()[-1])
    ks={k["kernel"]:(k["launches"],round(k["avg_us"],1)) for k in d["roofline"]["kernels"]}
    print("$name", "$*", round(d["value"],2), round(d["ms_per_step"],1), {k:v for k,v in ks.items() if "qr" in k or "apply" in k})
except Exception as e: print("$name ERR", e)
PY
}
run b128_s4 128 TRX_QR_SUPER=4
run b128_s4_nopipe 128 TRX_QR_SUPER=4 TRX_SLAB_PIPE=1
run b128_s8 128 TRX_QR_SUPER=8
run b128_s2 128 TRX_QR_SUPER=2
run b128_s4_g2 128 TRX_QR_SUPER=4 TRX_QR_GROUPS=2
run b128_s4_g8 128 TRX_QR_SUPER=4 TRX_QR_GROUPS=8
run b16_s4 16 TRX_QR_SUPER=4
run b16_s8 16 TRX_QR_SUPER=8
run b16_s4_aed64 16 TRX_QR_SUPER=4 TRX_QR_AED=64
run b16_s4_aed32 16 TRX_QR_SUPER=4 TRX_QR_AED=32
tail -3 gpurun_out/r5c_eigtests.txt
