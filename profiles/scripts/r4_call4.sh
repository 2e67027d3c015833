#!/bin/bash
# Round 4, fourth GPU call: where does the large-tile GEMM lose the matrix pipe?  Diagnostic modes (cache-resident operands, no loads), counters.
R=$GRAFT_REPO_ROOT
cd $R
export TRX_BENCH_NOPROF=1
g() { timeout 120 python tests/gpu_gemm_pmc.py 1922 1922 1922 128 2>&1 | grep TFLOP | sed "s/^/  $1 /"; }
for cfg in 1 2 4; do
  for dbg in 0 1 2 3; do
    TRX_GEMM_BIG=$cfg TRX_GEMM_BIG_DBG=$dbg g "cfg=$cfg dbg=$dbg"
  done
done
echo "== counters, cfg 2"
TRX_GEMM_BIG=2 bash profiles/scripts/pmc_gemm_hot.sh big2
echo "== counters, cfg 2, cache-resident operands (dbg 1)"
TRX_GEMM_BIG=2 TRX_GEMM_BIG_DBG=1 bash profiles/scripts/pmc_gemm_hot.sh big2d1 2>&1 | grep -v "FETCH\|WRITE"
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
for cfg in 0 2; do
  echo -n "bench batch 128 TRX_GEMM_BIG=$cfg: "; TRX_GEMM_BIG=$cfg timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
done
TRX_GEMM_BIG=2 timeout 200 python tests/gpu_gemm_bench.py hot 2>&1 | grep -v amdgpu
