#!/bin/bash
# Round 4, call 16: large-tile GEMM with scalar-base direct loads (no vector address arithmetic for full slabs): correctness, rate, step.
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_blocks.py -q -m gpu -k "large_tile or test_gemm or lu_solve" 2>&1 | tail -2
for cfg in 0 2; do echo "== TRX_GEMM_BIG=$cfg"; TRX_GEMM_BIG=$cfg timeout 200 python tests/gpu_gemm_bench.py hot 2>&1 | grep -v amdgpu; done
export TRX_BENCH_NOPROF=1
for i in 1 2; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench batch 128:', round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms')"; done
