#!/bin/bash
# Round 4, sixth GPU call: the GPU suite with the large-tile GEMM as the default, and the cycle breakdown of the AED kernel (TRX_QR_DEBUG).
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -16
export TRX_BENCH_NOPROF=1
for b in 16 128; do
  echo "== TRX_QR_DEBUG, batch $b"
  TRX_QR_DEBUG=1 timeout 200 python bench.py --batch $b --steps 1 --warmup 1 --no-cpu-baseline --no-strong-leg 2>&1 | grep -v amdgpu | grep -i "libtrx\|qr\|aed\|schur\|window" | tail -14
done
