#!/bin/bash
# Round 6, call 22: paired launches of the Hessenberg column loop (TRX_HESS_PAIR=1 off / 0 automatic / 2 always) at several batch sizes; eig tests.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call22.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    ks={k['kernel']: (round(k['est_total_ms_per_step']), round(k['avg_us']), round(k['frac'],3)) for k in r['kernels'] if k['kernel'].startswith('hess')}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, ks)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call22.err | line >> $O; }
echo "== eig tests" >> $O
timeout 900 python -m pytest tests/test_eig.py -m gpu -q -x 2>&1 | tail -3 >> $O
for g in 1 0 1 0; do run TRX_HESS_PAIR=$g; done
for g in 1 2; do B=64 run TRX_HESS_PAIR=$g; done
for g in 1 2; do B=32 run TRX_HESS_PAIR=$g; done
for g in 1 2; do B=16 run TRX_HESS_PAIR=$g; done
for g in 1 0; do FLAGS="--precision native" run TRX_HESS_PAIR=$g; done
for g in 1 0; do FLAGS="--eig-route fp64" run TRX_HESS_PAIR=$g; done
for g in 1 0; do FLAGS="--config 3" B=64 run TRX_HESS_PAIR=$g; done
cat $O | cut -c1-500
