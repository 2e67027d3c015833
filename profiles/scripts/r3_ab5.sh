#!/bin/bash
# Round 3, GPU call: gemv rows-per-pass, LU row-split at batch 128, the new full-size tests, bench --config 3 / 5.
export TRX_BENCH_NOPROF=1
run() { echo -n "$* : "; env "$@" timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],2), round(d['ms_per_step'],1), d.get('numerical_failures'), d.get('hbm'))"; }
EXTRA=""
run X=0
run TRX_HESS_RPW=4
run TRX_LU_SPLIT_BATCH=128
run TRX_LU_SPLIT_BATCH=128 TRX_HESS_RPW=4
EXTRA="--batch 16"
run X=0
run TRX_HESS_RPW=4
run TRX_LU_SPLIT_BATCH=128
echo "== new full-size tests"
timeout 900 python -m pytest tests/test_fullsize_properties.py tests/test_aux_rows.py tests/test_eig.py -m gpu -x -q --durations=8 2>&1 | tail -16
echo "== bench --config 3 / 5"
unset TRX_BENCH_NOPROF
timeout 600 python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03_bench_config3.json; cut -c1-400 gpurun_out/r03_bench_config3.json
timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03_bench_config5.json; cut -c1-400 gpurun_out/r03_bench_config5.json
