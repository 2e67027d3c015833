#!/bin/bash
# Round 3: fp64 GEMM general tile through the direct-to-LDS ring vs the register-staged kernel
timeout 600 python -m pytest tests/test_blocks.py -m gpu -x -q -k gemm 2>&1 | tail -2
echo "== ring"; python tests/gpu_gemm_bench.py 2>&1 | grep -v amdgpu
echo "== register-staged"; TRX_GEMM_DMA=0 python tests/gpu_gemm_bench.py 2>&1 | grep -v amdgpu
export TRX_BENCH_NOPROF=1
run() { echo -n "$* : "; env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],3), round(d['ms_per_step'],1), d.get('numerical_failures'))"; }
EXTRA=""; run TRX_GEMM_DMA=1; run TRX_GEMM_DMA=0
