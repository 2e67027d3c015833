#!/bin/bash
# Round 3, GPU call 2: inverse-iteration eigenvector route (eigenvalues-only QR) vs Schur vectors: correctness + A/B.
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],2), round(d['ms_per_step'],1), d.get('numerical_failures'), d.get('parity_sample'))
r=d.get('roofline') or {}
for k in r.get('kernels',[]): print('    %-32s launches %7d avg_us %10.1f ms/step %8.1f frac %.3f' % (k['kernel'], k['launches'], k['avg_us'], k['est_total_ms_per_step'], k.get('frac',0)))
"; }
echo "== correctness"
timeout 900 python -m pytest tests/test_eig.py tests/test_pipeline.py tests/test_fullsize_golden.py -m gpu -x -q 2>&1 | tail -5
echo "== batch 128"
EXTRA=""
run TRX_EIG_VEC=1
run TRX_EIG_VEC=2
run TRX_EIG_VEC=2 TRX_QR_GROUPS=8
run TRX_EIG_VEC=2 TRX_QR_GROUPS=2
run TRX_EIG_VEC=2 TRX_SLAB_DYN=1
run TRX_EIG_VEC=2 TRX_SLAB_WGS=256
export TRX_BENCH_NOPROF=1
run TRX_EIG_VEC=1
run TRX_EIG_VEC=2
echo "== batch 16"
EXTRA="--batch 16"
run TRX_EIG_VEC=1
run TRX_EIG_VEC=2
run TRX_EIG_VEC=2 TRX_QR_GROUPS=4
