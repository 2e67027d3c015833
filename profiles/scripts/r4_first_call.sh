#!/bin/bash
# First GPU call of round 4 (≈8 min): what round 3 could not take at its final sources, then the knob sweep of the QR phase in the regime
# the mixed-precision route created (fp32 first stage: the QR phase is latency-bound, its four iteration groups overlap only 1.6-fold).
#   usage: bash profiles/scripts/r4_first_call.sh > gpurun_out/r4_first_call.txt 2>&1
R=$GRAFT_REPO_ROOT
cd $R
# the micro-benchmarks (built here if the in-tree binaries did not travel; same image, hipcc present)
mkdir -p tests/micro/_build
for m in group_loop mfma_ladder; do [ -x tests/micro/_build/$m ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tests/micro/_build/$m tests/micro/$m.hip 2>/dev/null; done
# 1. HBM counters of the default bench command (two passes, one counter each; never combined with other trace domains)
timeout 600 bash profiles/scripts/pmc_bench.sh 128 $R/gpurun_out/r04_pmc_bench.json > gpurun_out/r04_pmc_bench.txt 2>&1
tail -20 gpurun_out/r04_pmc_bench.txt
cd $R
# 2. kernel trace with the per-stream lane view: UNDER THE TRACER every iteration group of round 3 was busy only 40 % of the QR phase, in idle periods of
#    more than 5 ms, and no two sweeps and no two AEDs of different groups ever ran together (profiles/r03_bench_final_concurrency.txt) --
#    the lanes show whether the groups pair up on shared hardware queues (s0/s2, s1/s3) or starve on the host
timeout 300 bash profiles/scripts/trace_bench.sh r04_first
head -12 gpurun_out/r04_first_lanes.txt | cut -c1-260
# 2b. the host side of the same question: HIP runtime API trace of one step (no counters)
timeout 300 bash profiles/scripts/api_trace.sh r04_first
cd $R
# 2c. the host loop of the QR phase with spin kernels instead of the real ones (tests/micro/group_loop.hip, built in-tree before the call:
#     hipcc --offload-arch=gfx950 -O2 -o tests/micro/_build/group_loop tests/micro/group_loop.hip): 1 / 2 / 4 groups, slab width, null
#     stream or not, with and without the memset + memcpy of the summary
for cfg in "4 512 1 0" "8 512 0 0"; do tests/micro/_build/group_loop $cfg; done
GPU_MAX_HW_QUEUES=8 tests/micro/_build/group_loop 4 512 1 0
GPU_MAX_HW_QUEUES=8 tests/micro/_build/group_loop 8 512 0 0
# 2e. GEMM shapes of the hot path (branch r4-prep: prefetch fix in the kernel; knob gemm_xcd = XCD-aware tile order)
python tests/gpu_gemm_bench.py 2>&1 | grep -v amdgpu
TRX_GEMM_XCD=1 python tests/gpu_gemm_bench.py 2>&1 | grep -v amdgpu
# 2d. where the fp64 GEMM loses the matrix pipe (0.59 issued): the ladder from a register-only MFMA loop to the full slab staging
#     (hipcc --offload-arch=gfx950 -O3 -o tests/micro/_build/mfma_ladder tests/micro/mfma_ladder.hip before the call)
timeout 120 tests/micro/_build/mfma_ladder
export TRX_BENCH_NOPROF=1
run() { echo -n "$* : "; env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],3), round(d['ms_per_step'],1), d.get('numerical_failures'))"; }
echo "== batch 128, mixed route: slab launch width (co-residency of the chase / AED workgroups of the other groups), groups, AED window"
EXTRA=""
run X=0
run TRX_GEMM_XCD=1
# wave priority of the chase / AED kernels (branch r4-prep)
run TRX_QR_PRIO=3
run TRX_QR_PRIO=1
run TRX_QR_PRIO=3 TRX_SLAB_WGS=256
# MORE slab workgroups: the fp32 slab kernel needs 108 VGPRs and 35 KB of LDS, so four workgroups fit on a CU where the default grid
# (512) places two; per wave a strip is claim (atomic with return, ~1.5 us exposed) + 16 loads (~2 us exposed) + MFMAs + stores
run TRX_SLAB_WGS=1024
run TRX_SLAB_WGS=768
run TRX_SLAB_WGS=1024 TRX_SLAB_SPW=2
run TRX_SLAB_WGS=384
run TRX_SLAB_WGS=256
run TRX_SLAB_WGS=192
run TRX_SLAB_WGS=128
run TRX_QR_GROUPS=8
run TRX_QR_GROUPS=8 TRX_SLAB_WGS=128
run TRX_QR_GROUPS=2
run TRX_QR_AED=48
run TRX_SLAB_DYN=1
# hardware queues of the process (read by the HIP runtime at start-up): 4 by default, streams beyond that share a queue and serialise
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=8 TRX_QR_GROUPS=8
run GPU_MAX_HW_QUEUES=2
# the step on a created stream instead of the null stream (shifts which hardware queue group 0 shares)
run TRX_BENCH_SIDE_STREAM=1
run TRX_BENCH_SIDE_STREAM=1 GPU_MAX_HW_QUEUES=8
# two / three bulge chains per sweep: a third fewer outer iterations (= AEDs) for a third more slab work, which is cheap in fp32
run TRX_QR_CHAINS=2
run TRX_QR_CHAINS=3
run TRX_QR_CHAINS=2 TRX_SLAB_WGS=256
echo "== two half-batches on two streams (host threads): the latency-bound QR phase of one under the GEMM phases of the other"
EXTRA="--streams 2"
run X=0
run TRX_QR_GROUPS=2
echo "== batch 16"
EXTRA="--batch 16"
run X=0
run TRX_QR_GROUPS=1
run TRX_QR_GROUPS=4
run TRX_QR_AED=64
run TRX_QR_CHAINS=2
