#!/bin/bash
# Round 4, call 18: lu_split_final_kernel with wave-parallel bookkeeping and staged row moves: LU tests, batch 16 and config 5.
R=$GRAFT_REPO_ROOT
cd $R
timeout 200 python -m pytest tests/test_blocks.py -q -m gpu -k "lu_" 2>&1 | tail -2
export TRX_BENCH_NOPROF=1
timeout 200 python bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 16:', round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))"
timeout 200 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5:', round(d['ms_per_step'],1), 'ms', d.get('fom'), d.get('grad_norm'))"
