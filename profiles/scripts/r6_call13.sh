#!/bin/bash
# Round 6, thirteenth GPU call: XCD-aware tile order of the GEMM kernels (TRX_GEMM_XCD=1 = plain blockIdx order), gemm_big on single full tile rows.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call13.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    po={p['phase'].split(' ')[0]: round(p['ms_per_step']) for p in r['phases']['phases']}
    ks={k['kernel']: (round(k['est_total_ms_per_step']), round(k['frac'],3)) for k in r['kernels'] if k['kernel'].startswith(('gemm','apply'))}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, po, ks)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call13.err | line >> $O; }
echo "== gemm / lu tests" >> $O
timeout 900 python -m pytest tests/test_blocks.py -m gpu -q -x 2>&1 | tail -3 >> $O
for g in 1 0 1 0; do run TRX_GEMM_XCD=$g; done
for g in 1 0; do B=16 run TRX_GEMM_XCD=$g; done
for g in 1 0; do FLAGS="--precision native" run TRX_GEMM_XCD=$g; done
for g in 1 0; do FLAGS="--config 5" B=1 run TRX_GEMM_XCD=$g; done
for g in 1 0; do FLAGS="--config 3" B=64 run TRX_GEMM_XCD=$g; done
cat $O | cut -c1-600
