#!/bin/bash
export TRX_BENCH_NOPROF=1
run() { echo -n "$* : "; env "$@" timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>&1 | grep -v amdgpu | python -c "
import sys,json
txt=sys.stdin.read().strip().splitlines()
for l in txt:
    if 'libtrx eig_refine: n' in l: print('   ', l[20:150])
d=json.loads(txt[-1])
print(round(d['value'],3), round(d['ms_per_step'],1), d.get('numerical_failures'), d.get('txx00_sample'))"; }
export TRX_EIG_DEBUG=1
EXTRA=""; run TRX_QR_TOL32=1; run TRX_QR_TOL32=16; run TRX_QR_TOL32=256
