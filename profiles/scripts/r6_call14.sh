#!/bin/bash
# Round 6, fourteenth GPU call: 3M complex product in the in-kernel left update of the fused launches (TRX_QR_VAR=5, prebuilt) against the 4M form.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call14.txt
: > $O
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    ks={k['kernel']: (round(k['est_total_ms_per_step']), round(k['avg_us'])) for k in r['kernels'] if k['kernel'].startswith(('qr','apply'))}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, ks)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call14.err | line >> $O; }
for lib in /tmp/libtrx_tip.so profiles/_ab_libs/15_qrvar5.so /tmp/libtrx_tip.so profiles/_ab_libs/15_qrvar5.so; do
  cp $lib torcwa_amd/libtrx.so
  echo "-- $(basename $lib)" >> $O
  run X=lib
  B=64 run X=lib
  B=16 run X=lib
done
cp profiles/_ab_libs/15_qrvar5.so torcwa_amd/libtrx.so
echo "== eig tests on the 3M variant" >> $O
timeout 900 python -m pytest tests/test_eig.py -m gpu -q -x 2>&1 | tail -3 >> $O
FLAGS="--eig-route fp64" run X=var5
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
FLAGS="--eig-route fp64" run X=tip
cat $O | cut -c1-600
