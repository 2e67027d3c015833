#!/bin/bash
# Round 4, call 10: look-ahead LU on hardware -- correctness (bit-identical factors, full LU tests), effect on the step (batch 128 / 16) and on config 5.
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_blocks.py -q -m gpu -k "lu_ or inverse or hmodes or redheffer" 2>&1 | tail -3
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), d.get('fom'))
except Exception as e: print('FAILED', e)"; }
for look in 1 0 1 0; do
  echo -n "batch 128 TRX_LU_LOOK=$look (1 = off): "; TRX_LU_LOOK=$look timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
done
for look in 1 0; do
  echo -n "batch 16  TRX_LU_LOOK=$look: "; TRX_LU_LOOK=$look timeout 300 python bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
  echo -n "config 5  TRX_LU_LOOK=$look: "; TRX_LU_LOOK=$look timeout 300 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | line
done
