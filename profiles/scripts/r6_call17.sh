#!/bin/bash
# Round 6, seventeenth GPU call: chains x AED window x groups at small batches (follow-up of r6p).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call17.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call17.err | line >> $O; }
B=16 run TRX_QR_CHAINS=3 TRX_QR_AED=64
B=16 run TRX_QR_CHAINS=2 TRX_QR_AED=64 TRX_QR_GROUPS=4
B=16 run TRX_QR_CHAINS=3 TRX_QR_AED=64 TRX_QR_GROUPS=4
B=16 run TRX_QR_CHAINS=1 TRX_QR_AED=64
B=16 run TRX_QR_CHAINS=2 TRX_QR_AED=56
B=8 run TRX_QR_CHAINS=2 TRX_QR_AED=64
B=8 run TRX_QR_CHAINS=3 TRX_QR_AED=64
B=8 run TRX_QR_CHAINS=2 TRX_QR_AED=64 TRX_QR_GROUPS=4
B=24 run TRX_QR_CHAINS=2 TRX_QR_AED=64
B=32 run TRX_QR_CHAINS=2 TRX_QR_AED=64
B=32 run TRX_QR_CHAINS=1 TRX_QR_AED=64
B=48 run TRX_QR_CHAINS=2 TRX_QR_AED=64
B=4 run TRX_QR_CHAINS=1
B=4 run TRX_QR_CHAINS=2 TRX_QR_AED=64
B=4 run TRX_QR_CHAINS=3 TRX_QR_AED=64
cat $O | cut -c1-300
