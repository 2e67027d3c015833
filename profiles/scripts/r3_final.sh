#!/bin/bash
# Round-3 evidence, one GPU call: HBM counters, the default bench line, the kernel trace, the batch-size lines, the fp64-only route beside
# the mixed one, configs 5 and 3.   usage: bash profiles/scripts/r3_final.sh     (outputs under gpurun_out/r03_*)
TAG=r03
R=$GRAFT_REPO_ROOT
cd $R
timeout 400 bash profiles/scripts/pmc_bench.sh 128 $R/gpurun_out/${TAG}_pmc_bench.json > gpurun_out/${TAG}_pmc_bench.txt 2>&1
cp gpurun_out/${TAG}_pmc_bench.json profiles/${TAG}_pmc_bench.json
cd $R
timeout 300 bash profiles/scripts/trace_bench.sh ${TAG}_bench_final
cp gpurun_out/${TAG}_bench_final_kernel_profile.json profiles/${TAG}_kernel_profile.json
cd $R
timeout 400 python bench.py > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/${TAG}_bench_final.err; tail -c 400 gpurun_out/${TAG}_bench_final.json
for b in 16 32 64; do timeout 200 python bench.py --batch $b --steps 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_b$b.json; done
TRX_EIG_VEC=1 timeout 200 python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_schur_fp64.json
timeout 200 python bench.py --precision native --steps 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_native.json
timeout 200 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_config5.json
timeout 200 python bench.py --config 4 --points 512 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_config4.json
timeout 500 python bench.py --config 3 --steps 1 --warmup 0 --no-cpu-baseline 2>gpurun_out/${TAG}_bench_config3.err | grep '^{' > gpurun_out/${TAG}_bench_config3.json
[ -s gpurun_out/${TAG}_bench_config3.json ] || timeout 500 python bench.py --config 3 --steps 1 --warmup 0 --chunk 32 --no-cpu-baseline 2>>gpurun_out/${TAG}_bench_config3.err | grep '^{' > gpurun_out/${TAG}_bench_config3.json
for f in final b16 b32 b64 schur_fp64 native config5 config4 config3; do python - $f <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r03_bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"], 3), d["unit"], round(d["ms_per_step"], 1), "ms/step")
except Exception as e:
    print(f, "FAILED", e)
PY
done
tail -3 gpurun_out/${TAG}_bench_config3.err
