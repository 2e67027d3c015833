#!/bin/bash
# Round 6, call 23: fp32 BLAS-2 stream of the Hessenberg reduction with 16-byte loads (pairs of elements) against the 8-byte form (library before).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call23.txt
: > $O
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    ks={k['kernel']: (round(k['est_total_ms_per_step']), round(k['avg_us']), round(k['frac'],3)) for k in r['kernels'] if k['kernel'].startswith('hess')}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, ks)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call23.err | line >> $O; }
for lib in profiles/_ab_libs/00_head.so /tmp/libtrx_tip.so profiles/_ab_libs/00_head.so /tmp/libtrx_tip.so; do
  cp $lib torcwa_amd/libtrx.so
  echo "-- $(basename $lib)" >> $O
  run X=lib
  B=16 run X=lib
  B=64 run X=lib
done
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
FLAGS="--precision native" run X=tip
FLAGS="--config 3" B=64 run X=tip
TRX_HESS_RPW=4 run X=tip
echo "== eig tests" >> $O
timeout 900 python -m pytest tests/test_eig.py -m gpu -q -x 2>&1 | tail -3 >> $O
cat $O | cut -c1-500
