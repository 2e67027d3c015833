#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call9.txt
: > $O
timeout 600 python tests/gpu_lu_sub.py >> $O 2>>gpurun_out/r6_call9.err
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in d['roofline']['phases']['inside_trx_eig']} if d.get('roofline') and d['roofline'].get('phases') else {}
    po={p['phase'].split(' ')[0]: round(p['ms_per_step']) for p in d['roofline']['phases']['phases']} if d.get('roofline') and d['roofline'].get('phases') else {}
    kk={k['kernel']: (round(k['est_total_ms_per_step']), round(k['frac'],3)) for k in d['roofline']['kernels'] if 'fp32' in k['kernel']} if d.get('roofline') else {}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, po, kk)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call9.err | line >> $O; }
echo "== fp32 128 x 128 tile on / off (gemm_big = 4 switches both large tiles off: compare the fp32 rows)" >> $O
run X=tip
run X=tip
B=16 run X=tip
FLAGS="--precision native" run X=tip
timeout 300 python -m pytest tests/test_blocks.py -m gpu -q -k "gemm" 2>&1 | tail -4 >> $O
cat $O | cut -c1-500
