#!/bin/bash
# Round 6, sixth GPU call: the rotation generator's rescue path (cost in the QR phase), the GPU suite with full failure output.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call6.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in d['roofline']['phases']['inside_trx_eig']} if d.get('roofline') and d['roofline'].get('phases') else {}
    po={p['phase'].split(' ')[0]: round(p['ms_per_step']) for p in d['roofline']['phases']['phases']} if d.get('roofline') and d['roofline'].get('phases') else {}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, po)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call6.err | line >> $O; }
run X=tip
run X=tip
B=16 run X=tip
FLAGS="--config 4 --points 512" run X=auto
FLAGS="--config 3" B=64 run X=auto
echo "== test_eig, first failure in full" >> $O
timeout 900 python -m pytest tests/test_eig.py -m gpu -q -x 2>&1 | tail -150 >> $O
echo "== gpu tests without test_eig" >> $O
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_eig.py 2>&1 | tail -30 >> $O
cat $O | cut -c1-600
