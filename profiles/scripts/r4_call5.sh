#!/bin/bash
# Round 4, fifth GPU call: 8-wave large-tile GEMM (128 x 96, two waves per SIMD, accumulators in VGPRs) against the one-wave-per-SIMD layouts.
R=$GRAFT_REPO_ROOT
cd $R
export TRX_BENCH_NOPROF=1
timeout 600 python -m pytest tests/test_blocks.py -q -m gpu -k "large_tile" -x 2>&1 | tail -3
for cfg in 2 3; do
  echo "== TRX_GEMM_BIG=$cfg"
  TRX_GEMM_BIG=$cfg timeout 200 python tests/gpu_gemm_bench.py hot 2>&1 | grep -v amdgpu
done
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
for cfg in 3 0 3; do
  echo -n "bench batch 128 TRX_GEMM_BIG=$cfg: "; TRX_GEMM_BIG=$cfg timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
done
echo "== counters, cfg 3"
TRX_GEMM_BIG=3 bash profiles/scripts/pmc_gemm_hot.sh big3 2>&1 | grep -v "LDS\|GRBM"
