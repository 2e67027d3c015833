#!/bin/bash
# Round 3: fused Hessenberg panel (row-local work on the wide launch): correctness on the GPU + timing at batch 128 / 16 / config 5
timeout 900 python -m pytest tests/test_eig.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -3
run() { echo -n "$* : "; env "$@" timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],3), round(d['ms_per_step'],1), d.get('numerical_failures'))
r=d.get('roofline') or {}
for k in r.get('kernels',[]):
    if 'hess' in k['kernel']: print('    %-32s launches %7d avg_us %10.1f ms/step %8.1f frac %.3f' % (k['kernel'], k['launches'], k['avg_us'], k['est_total_ms_per_step'], k.get('frac',0)))
"; }
EXTRA=""; run X=0
EXTRA="--batch 16"; run X=0
EXTRA="--config 5"; run X=0
export TRX_BENCH_NOPROF=1
EXTRA=""; run X=0; run TRX_HESS_RPW=2
