#!/bin/bash
# Round 6, eighteenth GPU call: the new automatic chain count / AED window at every batch size, the repaired balancing, full GPU suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call18.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call18.err | line >> $O; }
for b in 4 8 16 32 48 64 128; do B=$b run X=auto; done
B=64 run TRX_QR_CHAINS=2
B=16 run X=auto --precision native
FLAGS="--precision native" B=16 run X=auto
FLAGS="--eig-route fp64" B=16 run X=auto
FLAGS="--config 5" B=1 run X=auto
echo "== gpu suite" >> $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 >> $O
cat $O | cut -c1-300
