#!/bin/bash
# occupancy experiments of the QR phase (knobs only, one library): lean window kernel / simple vs pipelined update kernels / LDS reserve
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* : "; env "$@" timeout 200 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line; }
run TRX_SLAB_PIPE=0
run TRX_SLAB_PIPE=1
run TRX_SLAB_PIPE=1 TRX_QR_REGS=1
run TRX_SLAB_PIPE=0 TRX_QR_REGS=1
run TRX_SLAB_PIPE=1 TRX_QR_REGS=1 TRX_SLAB_LDS=54
run TRX_SLAB_PIPE=1 TRX_QR_REGS=3 TRX_SLAB_LDS=54
run TRX_SLAB_PIPE=1 TRX_QR_REGS=1 TRX_QR_SUPER=8
run TRX_SLAB_PIPE=1 TRX_QR_REGS=1 TRX_QR_SUPER=2
run TRX_SLAB_PIPE=1 TRX_SLAB_SPW=1
run TRX_SLAB_PIPE=1 TRX_SLAB_SPW=4
B=16 run TRX_SLAB_PIPE=1
B=16 run TRX_SLAB_PIPE=1 TRX_QR_REGS=1
