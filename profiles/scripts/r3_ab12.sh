#!/bin/bash
# full gpu suite with the mixed-precision default + config 5 / config 3 timing
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -12
export TRX_BENCH_NOPROF=1
run() { echo -n "$* : "; env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],3), round(d['ms_per_step'],1), d.get('numerical_failures'), d.get('fom'), d.get('grad_norm'), (d.get('hbm') or {}).get('peak_allocated_GB'))"; }
EXTRA="--config 5"; run TRX_EIG_VEC=0; run TRX_EIG_VEC=1
EXTRA="--batch 64"; run TRX_EIG_VEC=0; run TRX_EIG_VEC=1
