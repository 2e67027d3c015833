#!/bin/bash
# Round 6, eleventh GPU call: triangular solves by recursive halving (lu.hip: tri_block_solve) against the leaf-by-leaf form (library of the
# commit before), bench bookkeeping (dominant kernel by wall share), LU / S-matrix / full-size parity tests.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call11.txt
: > $O
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    po={p['phase'].split(' ')[0]: round(p['ms_per_step']) for p in r['phases']['phases']}
    ks={k['kernel']: round(k['est_total_ms_per_step']) for k in r['kernels'] if k['kernel'].startswith('gemm')}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, po, ks, 'dominant', r['dominant_kernel']['kernel'], round(r['dominant_kernel']['frac'],3))
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call11.err | line >> $O; }
for lib in profiles/_ab_libs/00_head.so /tmp/libtrx_tip.so; do
  cp $lib torcwa_amd/libtrx.so
  echo "-- $(basename $lib)" >> $O
  run X=lib
  run X=lib
  B=16 run X=lib
  FLAGS="--precision native" run X=lib
  FLAGS="--config 5" B=1 run X=lib
done
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
echo "== gpu tests" >> $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 >> $O
cat $O | cut -c1-700
