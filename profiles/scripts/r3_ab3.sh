#!/bin/bash
# Round 3, GPU call 3: inverse-iteration kernel v3 (owner-published pivots) layouts + QR knobs in the eigenvalues-only regime.
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],2), round(d['ms_per_step'],1), d.get('numerical_failures'))
r=d.get('roofline') or {}
for k in r.get('kernels',[]):
    if k['kernel'] in ('invit_solve_kernel','qr_prepare_kernel','qr_window_kernel','apply_window_kernel'): print('    %-32s launches %7d avg_us %10.1f ms/step %8.1f' % (k['kernel'], k['launches'], k['avg_us'], k['est_total_ms_per_step']))
"; }
python - <<'PY'
import sys
sys.path.insert(0,'.')
PY
echo "== batch 128, layouts of the inverse-iteration kernel"
EXTRA=""
run TRX_INVIT_CFG=0
run TRX_INVIT_CFG=2
run TRX_INVIT_CFG=3
echo "== QR knobs, eigenvalues only"
run TRX_QR_CHAINS=2
run TRX_QR_CHAINS=3
run TRX_QR_CHAINS=3 TRX_QR_AED=48
run TRX_QR_AED=48
run TRX_QR_AED=32
run TRX_QR_CHAINS=2 TRX_QR_GROUPS=8
run TRX_SLAB_DYN=1 TRX_SLAB_SPW=1
run TRX_SLAB_WGS=256
run TRX_SLAB_WGS=128
echo "== batch 16"
EXTRA="--batch 16"
run X=0
run TRX_QR_CHAINS=3
run TRX_QR_CHAINS=3 TRX_QR_GROUPS=1
echo "== correctness of the new default"
timeout 900 python -m pytest tests/test_eig.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -3
