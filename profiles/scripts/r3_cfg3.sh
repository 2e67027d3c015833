#!/bin/bash
# config 3 (4-layer stack, order [21,21]) with the streaming cascade: chunk 32 vs 64
export TRX_BENCH_NOPROF=1
for c in 32 64; do
  timeout 900 python bench.py --config 3 --steps 1 --warmup 0 --chunk $c --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk', d['config']['chunk'], round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('hbm'), d.get('txx00_sample'))"
done
