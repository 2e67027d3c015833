#!/bin/bash
# Round 6, call 26: the AED's policy knobs on the round's final kernels (nibble percentage, reordering moves), batch 128 and 16.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call26.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call26.err | line >> $O; }
run X=auto
for v in 50 75; do run TRX_QR_NIBBLE=$v; done
for v in 6 20 32; do run TRX_QR_MOVES=$v; done
B=16 run X=auto
for v in 50 75; do B=16 run TRX_QR_NIBBLE=$v; done
for v in 6 20 32; do B=16 run TRX_QR_MOVES=$v; done
B=16 run TRX_QR_CHAINS=3
cat $O | cut -c1-260
