#!/bin/bash
# Round 4: bench lines of the other GPU configs at the library defaults (config 3: 4-layer stack at [21,21], 64 points in one chunk; config 4:
# 512-point sample of the (Wx, Wy, lambda) sweep in chunks of 256; config 5: forward + adjoint at [25,25]; native precision for reference).
R=$GRAFT_REPO_ROOT
cd $R
export TRX_BENCH_NOPROF=1
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1', round(d['value'],3), d['unit'], round(d['ms_per_step'],1), 'ms/step', d['config'].get('eig_route'), 'peak alloc GB', (d.get('hbm') or {}).get('peak_allocated_GB'), 'failures', d.get('numerical_failures'), d.get('fom'), d.get('grad_norm'))"; }
timeout 500 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_config3.json 2> gpurun_out/r04_bench_config3.err; show gpurun_out/r04_bench_config3.json; tail -2 gpurun_out/r04_bench_config3.err
timeout 300 python bench.py --config 4 --points 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_config4.json 2>/dev/null; show gpurun_out/r04_bench_config4.json
timeout 300 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_config5.json 2>/dev/null; show gpurun_out/r04_bench_config5.json
timeout 300 python bench.py --precision native --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_native.json 2>/dev/null; show gpurun_out/r04_bench_native.json
