#!/bin/bash
# Round 4, third GPU call: the large-tile fp64 GEMM (gemm_big.hip) -- correctness on hardware, rate per tile configuration, effect on the step.
R=$GRAFT_REPO_ROOT
cd $R
export TRX_BENCH_NOPROF=1
timeout 600 python -m pytest tests/test_blocks.py -q -m gpu -k "large_tile or test_gemm" -x 2>&1 | tail -5
for cfg in 0 1 2 3 4; do
  echo "== TRX_GEMM_BIG=$cfg"
  TRX_GEMM_BIG=$cfg timeout 200 python tests/gpu_gemm_bench.py hot 2>&1 | grep -v amdgpu
done
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), d.get('parity_sample'))
except Exception as e: print('FAILED', e)"; }
for cfg in 0 1 2; do
  echo -n "bench batch 128 TRX_GEMM_BIG=$cfg: "; TRX_GEMM_BIG=$cfg timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
done
