#!/bin/bash
# Round 6, seventh GPU call: sub-blocked LU panels (row-split panel in sub-blocks of 8 columns, one-workgroup panel with LDS-resident
# sub-blocks) against the library before them, the rotation generator's rescue branch behind the fast path, GPU suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call7.txt
: > $O
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in d['roofline']['phases']['inside_trx_eig']} if d.get('roofline') and d['roofline'].get('phases') else {}
    po={p['phase'].split(' ')[0]: round(p['ms_per_step']) for p in d['roofline']['phases']['phases']} if d.get('roofline') and d['roofline'].get('phases') else {}
    lu=[round(k['est_total_ms_per_step']) for k in d['roofline']['kernels'] if k['kernel']=='lu_panel_kernel'] if d.get('roofline') else []
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, po, 'lu_panel', lu)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call7.err | line >> $O; }
for lib in profiles/_ab_libs/*.so /tmp/libtrx_tip.so; do
  cp $lib torcwa_amd/libtrx.so
  echo "-- $(basename $lib)" >> $O
  run X=lib
  run X=lib
  B=16 run X=lib
  FLAGS="--config 5" B=1 run X=lib
done
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
echo "== gpu tests" >> $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 >> $O
cat $O | cut -c1-600
