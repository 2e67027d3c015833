#!/bin/bash
# Round 6, fourth GPU call: forwarding alone (TRX_QR_FWD), the incremental balancing sweep and the tiled coupling-graph kernel, GPU suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call4.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in d['roofline']['phases']['inside_trx_eig']} if d.get('roofline') and d['roofline'].get('phases') else {}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call4.err | line >> $O; }
run X=tip
run TRX_QR_FWD=1
run X=tip
run TRX_QR_FWD=1
B=16 run X=tip
B=16 run TRX_QR_FWD=1
B=16 run X=tip
B=16 run TRX_QR_FWD=1
echo "== gpu tests" >> $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 >> $O
cat $O | cut -c1-400
