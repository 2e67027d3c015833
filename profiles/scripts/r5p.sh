#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
(timeout 600 python -m pytest tests/test_blocks.py tests/test_eig.py -m gpu -x -q 2>&1 | tail -2)
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), d.get('fom'))
except Exception as e: print('FAILED', e)"; }
echo -n "batch 128: "; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line
echo -n "batch 16: "; timeout 300 python bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | line
echo -n "config 5: "; timeout 300 python bench.py --config 5 --steps 2 --warmup 1 2>/dev/null | line
echo -n "config 4 auto: "; timeout 600 python bench.py --config 4 --points 512 --steps 2 --warmup 1 2>/dev/null | line
echo -n "config 4 fp64 chunk 256: "; timeout 600 python bench.py --config 4 --points 512 --chunk 256 --eig-route fp64 --steps 2 --warmup 1 2>/dev/null | line
echo -n "config 4 auto chunk 256: "; timeout 600 python bench.py --config 4 --points 512 --chunk 256 --steps 2 --warmup 1 2>/dev/null | line
echo -n "config 3 auto: "; timeout 900 python bench.py --config 3 --steps 1 --warmup 1 2>/dev/null | line
