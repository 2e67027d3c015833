#!/bin/bash
# Round 6, fifth GPU call: fused launches (chase + far part of the previous launch's left update) A/B, the two failing eig tests with details.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call5.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in d['roofline']['phases']['inside_trx_eig']} if d.get('roofline') and d['roofline'].get('phases') else {}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, d.get('txx00_sample'))
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call5.err | line >> $O; }
run X=fused
run TRX_QR_FUSE=1
run X=fused
run TRX_QR_FUSE=1
run TRX_QR_SUPER=8
run TRX_QR_SUPER=2
run TRX_QR_GROUPS=8
B=16 run X=fused
B=16 run TRX_QR_FUSE=1
B=16 run TRX_QR_SUPER=8
B=16 run TRX_QR_GROUPS=4
B=64 run X=fused
B=32 run X=fused
FLAGS="--eig-route fp64" run X=fused
FLAGS="--eig-route fp64" run TRX_QR_FUSE=1
FLAGS="--config 5" B=1 run X=fused
echo "== eig tests" >> $O
timeout 900 python -m pytest tests/test_eig.py -m gpu -q -k "mixed_precision_route or partial_fallback" 2>&1 | grep -v "^  \|^$" | tail -60 >> $O
timeout 900 python -m pytest tests/test_eig.py -m gpu -q 2>&1 | tail -5 >> $O
cat $O | cut -c1-400
