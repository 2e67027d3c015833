#!/bin/bash
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* : "; env "$@" timeout 200 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line; }
(timeout 300 python -m pytest tests/test_eig.py -m gpu -x -q 2>&1 | tail -2)
run X=0
run TRX_QR_SUPER=8
run TRX_QR_SUPER=6
run TRX_QR_GROUPS=3
run TRX_QR_GROUPS=6
B=16 run X=0
B=16 run TRX_QR_SUPER=8
B=16 run TRX_QR_GROUPS=1
B=32 run X=0
B=64 run X=0
unset TRX_BENCH_NOPROF
bash profiles/scripts/pmc_qr_updates.sh 128 2>&1 | grep -A20 "apply_links" | grep "==\|SQ_VALU_MFMA_BUSY\|SQ_BUSY_CYCLES\|WAIT_ANY\|FETCH\|WRITE"
