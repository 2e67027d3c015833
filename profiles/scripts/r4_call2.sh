#!/bin/bash
# Round 4, second GPU call: MFMA ladder, counter passes of the hot GEMM shape, counter passes of the whole bench (HEAD sources).
R=$GRAFT_REPO_ROOT
cd $R
[ -x tests/micro/_build/mfma_ladder ] || (mkdir -p tests/micro/_build && hipcc --offload-arch=gfx950 -O3 -o tests/micro/_build/mfma_ladder tests/micro/mfma_ladder.hip)
timeout 120 tests/micro/_build/mfma_ladder > gpurun_out/r04_mfma_ladder.txt 2>&1
(cd /tmp; rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1)
bash profiles/scripts/pmc_gemm_hot.sh main > gpurun_out/r04_gemm_pmc.txt 2>&1
sed -i 's/^  rocprofv3 --kernel-trace --pmc/  timeout 420 rocprofv3 --kernel-trace --pmc/' profiles/scripts/pmc_bench.sh
bash profiles/scripts/pmc_bench.sh 128 $R/gpurun_out/r04_pmc_bench.json > gpurun_out/r04_pmc_bench.txt 2>&1
cat gpurun_out/r04_mfma_ladder.txt gpurun_out/r04_gemm_pmc.txt; head -12 gpurun_out/r04_pmc_bench.txt
