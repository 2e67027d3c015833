#!/bin/bash
# Round 6, call 24: iteration groups of the QR phase at batch 128 with the fused launches (3 / 4 / 5 / 6 / 8), rows per wave of the fp32 stream.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call24.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call24.err | line >> $O; }
run X=auto
for g in 3 5 6 8; do run TRX_QR_GROUPS=$g; done
run TRX_QR_GROUPS=6 TRX_QR_SUPER=8
run TRX_QR_GROUPS=8 TRX_QR_SUPER=2
run TRX_HESS_RPW=2
B=64 run X=auto
B=64 run TRX_QR_GROUPS=6
B=64 run TRX_QR_GROUPS=8
cat $O | cut -c1-300
