#!/bin/bash
# Round 6, call 27: several chains per sweep -- the right / Z update of a window step riding in the next step's chase launch (automatic) against
# its own launch (TRX_QR_FUSE=1), batch 4 ... 48 and config 5; eig tests.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call27.txt
: > $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    ks={k['kernel']: (round(k['est_total_ms_per_step']), round(k['avg_us'])) for k in r['kernels'] if k['kernel'].startswith(('qr','apply'))}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, ks)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call27.err | line >> $O; }
echo "== eig tests" >> $O
timeout 900 python -m pytest tests/test_eig.py -m gpu -q -x 2>&1 | tail -3 >> $O
for b in 16 8 32 48 4; do
  B=$b run TRX_QR_FUSE=1
  B=$b run X=auto
done
B=16 run TRX_QR_FUSE=1
B=16 run X=auto
FLAGS="--config 5" B=1 run TRX_QR_FUSE=1
FLAGS="--config 5" B=1 run X=auto
echo "== full-size parity" >> $O
timeout 900 python -m pytest tests/test_fullsize_golden.py tests/test_pipeline.py tests/test_grad.py -m gpu -q -x 2>&1 | tail -3 >> $O
cat $O | cut -c1-420
