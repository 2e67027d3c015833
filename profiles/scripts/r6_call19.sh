#!/bin/bash
# Round 6, nineteenth GPU call: LDS-resident LU panels wherever they fit (sub-blocks of 8 / 4 columns) against the row-split panels above
# 1024 rows (library of the commit before); LU tests.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call19.txt
: > $O
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in r['phases']['inside_trx_eig']}
    po={p['phase'].split(' ')[0]: round(p['ms_per_step']) for p in r['phases']['phases']}
    lu=[round(k['est_total_ms_per_step']) for k in r['kernels'] if k['kernel']=='lu_panel_kernel']
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph, po, 'lu_panel', lu)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call19.err | line >> $O; }
echo "== lu tests (tip)" >> $O
timeout 900 python -m pytest tests/test_blocks.py -m gpu -q -x -k lu 2>&1 | tail -3 >> $O
for lib in profiles/_ab_libs/00_head.so /tmp/libtrx_tip.so profiles/_ab_libs/00_head.so /tmp/libtrx_tip.so; do
  cp $lib torcwa_amd/libtrx.so
  echo "-- $(basename $lib)" >> $O
  run X=lib
  B=16 run X=lib
  B=64 run X=lib
done
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
run TRX_LU_SUB=2
FLAGS="--precision native" run X=tip
FLAGS="--config 3" B=64 run X=tip
FLAGS="--config 5" B=1 run X=tip
FLAGS="--config 4 --points 512" run X=tip
cp profiles/_ab_libs/00_head.so torcwa_amd/libtrx.so
FLAGS="--precision native" run X=head
FLAGS="--config 3" B=64 run X=head
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
cat $O | cut -c1-500
