#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), d.get('fom'), d.get('grad_norm'))
except Exception as e: print('FAILED', e)"; }
run5() { echo -n "config 5 $* : "; env "$@" timeout 300 python bench.py --config 5 --steps 2 --warmup 1 2>/dev/null | line; }
run5 X=0
run5 TRX_QR_CHAINS=1
run5 TRX_QR_CHAINS=1 TRX_QR_SUPER=8
run5 TRX_QR_CHAINS=1 TRX_QR_SUPER=1
run5 TRX_QR_CHAINS=2
unset TRX_BENCH_NOPROF
bash profiles/scripts/r5_configs.sh
