#!/bin/bash
# cluster statistics of the mixed route on the geometry sweep (config 4) and the [21,21] stack (config 3): TRX_EIG_DEBUG of the first eig calls
cd $GRAFT_REPO_ROOT
export TRX_BENCH_NOPROF=1 TRX_EIG_DEBUG=1
timeout 300 python bench.py --config 4 --points 128 --chunk 128 --steps 1 --warmup 0 2>&1 >/dev/null | grep "eig_refine" | grep -v "indices in pairs" | head -12
timeout 600 python bench.py --config 3 --batch 16 --steps 1 --warmup 0 2>&1 >/dev/null | grep "eig_refine" | grep -v "indices in pairs" | head -24
