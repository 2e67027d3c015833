#!/bin/bash
export TRX_BENCH_NOPROF=1
run() { echo -n "$* : "; env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],3), round(d['ms_per_step'],1), d.get('numerical_failures'))"; }
EXTRA=""; run TRX_HESS_RPW=4; run TRX_HESS_RPW=2
EXTRA="--batch 16"; run TRX_HESS_RPW=4; run TRX_HESS_RPW=2
