#!/bin/bash
# precision="native" (fp32 arithmetic for the complex64 problem: the reference's own arithmetic class) on the round-5 sources
cd $GRAFT_REPO_ROOT
timeout 150 python bench.py --precision native --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r05_bench_native.json 2> gpurun_out/r05_bench_native.err
cut -c1-300 gpurun_out/r05_bench_native.json
timeout 80 python bench.py --precision native --batch 16 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r05_bench_native_b16.json 2>> gpurun_out/r05_bench_native.err
cut -c1-300 gpurun_out/r05_bench_native_b16.json
