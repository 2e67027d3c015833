#!/bin/bash
# Round 6, second GPU call: the CU-partitioned two-half-batch pipeline, LU sub-batches A/B, a kernel trace of the
# default step (refinement breakdown), the rest of the GPU suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call2.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in d['roofline']['phases']['inside_trx_eig']} if d.get('roofline') and d['roofline'].get('phases') else {}
    po={p['phase'].split(' ')[0]: round(p['ms_per_step']) for p in d['roofline']['phases']['phases']} if d.get('roofline') and d['roofline'].get('phases') else {}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, po)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call2.err | line >> $O; }
echo "== mixed route: residuals by Newton steps (n = 600)" >> $O
timeout 300 python tests/gpu_mixed_steps.py >> $O 2>>gpurun_out/r6_call2.err
echo "== LU sub-batches" >> $O
run X=default
run TRX_LU_PARTS=1
run TRX_LU_PARTS=3
B=16 run X=default
B=16 run TRX_LU_PARTS=1
echo "== two half-batches on two host threads: plain streams, then CU-masked streams (reserve K CUs, K / 8 per XCD)" >> $O
export TRX_LU_PARTS=1     # the side streams of the LU sub-batches are not CU-masked: off for the partition experiment
FLAGS="--streams 2" run X=plain
for K in 16 32 48 64; do
  FLAGS="--streams 2 --cu-reserve $K" run X=free
  FLAGS="--streams 2 --cu-reserve $K --cu-lanes reserved" run X=reserved
done
FLAGS="--streams 4 --cu-reserve 32" run X=free
FLAGS="--streams 4 --cu-reserve 64" run X=free
B=256 FLAGS="--streams 2 --cu-reserve 32" run X=free
B=256 run X=default
B=16 FLAGS="--streams 2 --cu-reserve 32" run X=free
unset TRX_LU_PARTS
echo "== kernel trace of the default step" >> $O
bash profiles/scripts/trace_bench.sh r6_call2_trace >> $O 2>&1
head -60 gpurun_out/r6_call2_trace_kernel_stats.txt >> $O
echo "== gpu tests (all failures)" >> $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 >> $O
cat $O
