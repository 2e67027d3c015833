#!/bin/bash
# Round 4, call 8: full GPU suite at the new defaults; in-LDS Schur solver: rotation broadcast by v_readlane + merged H / U right phase
# (default) against the round-3 code (TRX_QR_ROTB=1): cycle breakdown (TRX_QR_DEBUG) and step times.
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
for rb in 1 0; do
  echo "== TRX_QR_ROTB=$rb (1 = round-3 code)"
  TRX_QR_ROTB=$rb TRX_QR_DEBUG=1 timeout 200 python bench.py --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-strong-leg 2>&1 | grep "libtrx qr_prepare" | tail -1 | cut -c1-330
  echo -n "  batch 16 : "; TRX_QR_ROTB=$rb timeout 200 python bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
  echo -n "  batch 128: "; TRX_QR_ROTB=$rb timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
done
