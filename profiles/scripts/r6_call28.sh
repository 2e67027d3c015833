#!/bin/bash
# Round 6, call 28: strips per wave of the riding right / Z update (TRX_QR_RSPW = 1 / 2 / 4) with riding forced on (TRX_QR_FUSE=2) at batch 16 ... 48.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call28.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    ks={k['kernel']: (round(k['est_total_ms_per_step']), round(k['avg_us'])) for k in r['kernels'] if k['kernel'].startswith(('qr','apply'))}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, ks)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call28.err | line >> $O; }
for b in 32 48 24; do
  B=$b run TRX_QR_FUSE=1
  for r in 1 2 4; do B=$b run TRX_QR_FUSE=2 TRX_QR_RSPW=$r; done
done
for r in 1 2 4; do B=16 run TRX_QR_RSPW=$r; done
for r in 2 4; do B=8 run TRX_QR_RSPW=$r; done
for r in 2 4; do FLAGS="--config 5" B=1 run TRX_QR_RSPW=$r; done
cat $O | cut -c1-400
