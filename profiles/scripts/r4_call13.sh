#!/bin/bash
# Round 4, call 13: why are configs 3 and 4 slow on the default (mixed) route?  Refinement flag statistics (TRX_EIG_DEBUG) and the all-fp64 route.
R=$GRAFT_REPO_ROOT
cd $R
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
echo "== config 4, 256 points in one chunk, default route, TRX_EIG_DEBUG"
TRX_EIG_DEBUG=1 timeout 300 python bench.py --config 4 --points 256 --steps 1 --warmup 0 --no-cpu-baseline 2> gpurun_out/c4_dbg.err | line; grep "eig_refine" gpurun_out/c4_dbg.err | cut -c1-400 | tail -6
echo "== config 4, all-fp64 route"
TRX_EIG_VEC=1 timeout 300 python bench.py --config 4 --points 256 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | line
echo "== config 4, chunk 128, default route"
timeout 300 python bench.py --config 4 --points 256 --chunk 128 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | line
echo "== config 3, all-fp64 route"
TRX_EIG_VEC=1 timeout 400 python bench.py --config 3 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | line
echo "== config 3, default route, TRX_EIG_DEBUG"
TRX_EIG_DEBUG=1 timeout 400 python bench.py --config 3 --steps 1 --warmup 0 --no-cpu-baseline 2> gpurun_out/c3_dbg.err | line; grep "eig_refine" gpurun_out/c3_dbg.err | cut -c1-400 | tail -8
