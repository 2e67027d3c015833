#!/bin/bash
# Round 4, call 11: QR-phase knobs in the present regime (slab launch width, iteration groups, two host threads); kernel trace at batch 16.
R=$GRAFT_REPO_ROOT
cd $R
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$1: "; env $1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg $2 2>/dev/null | line; }
run "TRX_NONE=0" ""
run "TRX_SLAB_WGS=256" ""
run "TRX_SLAB_WGS=384" ""
run "TRX_SLAB_WGS=768" ""
run "TRX_SLAB_WGS=1024" ""
run "TRX_QR_GROUPS=2" ""
run "TRX_QR_GROUPS=6" ""
run "TRX_QR_GROUPS=8" ""
run "TRX_NONE=0" "--streams 2"
run "TRX_QR_AED=48" ""
run "TRX_NONE=0" ""
unset TRX_BENCH_NOPROF
bash profiles/scripts/trace_bench.sh r04_b16 --batch 16
head -24 gpurun_out/r04_b16_kernel_stats.txt; cat gpurun_out/r04_b16_phases.txt
