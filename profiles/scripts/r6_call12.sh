#!/bin/bash
# Round 6, twelfth GPU call: delayed right updates of the Hessenberg reduction (groups of 1 / 2 / 4 panels: TRX_HESS_GROUP), eigensolver tests.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call12.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    po={p['phase'].split(' ')[0]: round(p['ms_per_step']) for p in r['phases']['phases']}
    ks={k['kernel']: round(k['est_total_ms_per_step']) for k in r['kernels'] if k['kernel'].startswith('gemm')}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, po, ks)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call12.err | line >> $O; }
echo "== eig tests" >> $O
timeout 900 python -m pytest tests/test_eig.py -m gpu -q -x 2>&1 | tail -4 >> $O
for g in 1 4 2 1 4; do run TRX_HESS_GROUP=$g; done
for g in 1 4; do B=16 run TRX_HESS_GROUP=$g; done
for g in 1 4; do FLAGS="--precision native" run TRX_HESS_GROUP=$g; done
for g in 1 4; do FLAGS="--config 5" B=1 run TRX_HESS_GROUP=$g; done
for g in 1 4; do FLAGS="--eig-route fp64" run TRX_HESS_GROUP=$g; done
echo "== full-size parity" >> $O
timeout 900 python -m pytest tests/test_fullsize_golden.py tests/test_pipeline.py -m gpu -q -x 2>&1 | tail -4 >> $O
cat $O | cut -c1-500
