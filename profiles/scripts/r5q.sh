#!/bin/bash
cd $GRAFT_REPO_ROOT
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* : "; env "$@" timeout 200 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>/dev/null | line; }
FLAGS="--eig-route fp64"
run X=fp64
run TRX_QR_SUPER=1
run TRX_QR_DEFER=1
run TRX_QR_SUPER=1 TRX_QR_DEFER=1
run TRX_QR_SUPER=2 TRX_QR_DEFER=1
run TRX_QR_SUPER=8
FLAGS=""
run X=mixed
run TRX_QR_DEFER=1
run TRX_QR_SUPER=2 TRX_QR_DEFER=1
B=16 run TRX_QR_DEFER=1
