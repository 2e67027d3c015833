#!/bin/bash
# Round 6, call 29: strips per wave of the far workgroups of the one-chain fused launches (TRX_QR_FSPW = 1 / 2 / 4) at batch 128 and 64.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call29.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    ks={k['kernel']: (round(k['est_total_ms_per_step']), round(k['avg_us'])) for k in r['kernels'] if k['kernel'].startswith(('qr','apply'))}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, ks)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call29.err | line >> $O; }
for r in 1 2 4 1 2; do run TRX_QR_FSPW=$r; done
for r in 1 2 4; do B=64 run TRX_QR_FSPW=$r; done
for r in 1 2; do FLAGS="--precision native" run TRX_QR_FSPW=$r; done
cat $O | cut -c1-400
