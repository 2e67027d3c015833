#!/bin/bash
# Everything the round's committed evidence comes from, in one GPU call.  usage: bash profiles/scripts/final_round.sh <tag>   (e.g. r02)
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputests.log 2>&1; tail -2 gpurun_out/${TAG}_gputests.log
# HBM counters first (bench.py only accepts a PMC summary taken on the same kernel sources)
bash profiles/scripts/pmc_bench.sh 128 $R/gpurun_out/${TAG}_pmc_bench.json > gpurun_out/${TAG}_pmc_bench.txt 2>&1
cp gpurun_out/${TAG}_pmc_bench.json profiles/${TAG}_pmc_bench.json
cd $R
python bench.py > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/${TAG}_bench_final.err; tail -c 600 gpurun_out/${TAG}_bench_final.json
bash profiles/scripts/trace_bench.sh ${TAG}_bench_final
cd $R
for b in 16 32 64; do python bench.py --batch $b --steps 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_b$b.json; done
python bench.py --precision native --steps 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_native.json
python tests/gpu_gemm_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_gemm_shapes.txt
timeout 300 python tests/gpu_config5.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_config5.log
timeout 400 python tests/gpu_config3.py 21 32 32 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_config3.log
