#!/bin/bash
run() { echo -n "$* : "; env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],3), round(d['ms_per_step'],1), d.get('numerical_failures'))
r=d.get('roofline') or {}
for k in r.get('kernels',[]):
    if 'gemv' in k['kernel']: print('    %-32s launches %7d avg_us %10.1f ms/step %8.1f frac %.3f' % (k['kernel'], k['launches'], k['avg_us'], k['est_total_ms_per_step'], k.get('frac',0)))
"; }
EXTRA=""; run TRX_HESS_UNR=1; run TRX_HESS_UNR=2; run TRX_HESS_UNR=1 TRX_HESS_RPW=4
