#!/bin/bash
# Round 3: mixed-precision eigensolver (fp32 eigendecomposition + Newton refinement in fp64) -- correctness on the GPU and the bench
timeout 900 python -m pytest tests/test_eig.py tests/test_pipeline.py tests/test_fullsize_golden.py -m gpu -x -q 2>&1 | tail -4
run() { echo -n "$* : "; env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],3), round(d['ms_per_step'],1), d.get('numerical_failures'), d.get('txx00_sample'), (d.get('hbm') or {}).get('peak_allocated_GB'))"; }
export TRX_BENCH_NOPROF=1
EXTRA=""; run TRX_EIG_VEC=0; run TRX_EIG_VEC=1
EXTRA="--batch 16"; run TRX_EIG_VEC=0; run TRX_EIG_VEC=1
unset TRX_BENCH_NOPROF
EXTRA="--cpu-points 1"; timeout 600 python bench.py --steps 2 --warmup 1 --cpu-points 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('with parity sample:', round(d['value'],3), d.get('parity_sample'))"
