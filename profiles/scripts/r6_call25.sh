#!/bin/bash
# Round 6, call 25: right solves of the layer S-matrix done as right solves (lu_solve_right) against the transposed-system route (library before).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call25.txt
: > $O
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    po={p['phase'].split(' ')[0]: round(p['ms_per_step']) for p in r['phases']['phases']}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), po, d['parity_sample']['max_rel_err_vs_c128_oracle'] if d.get('parity_sample') else None)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call25.err | line >> $O; }
echo "== parity tests (tip)" >> $O
timeout 1200 python -m pytest tests/test_pipeline.py tests/test_fullsize_golden.py tests/test_fullsize_properties.py tests/test_blocks.py -m gpu -q -x 2>&1 | tail -3 >> $O
for lib in profiles/_ab_libs/00_head.so /tmp/libtrx_tip.so profiles/_ab_libs/00_head.so /tmp/libtrx_tip.so; do
  cp $lib torcwa_amd/libtrx.so
  echo "-- $(basename $lib)" >> $O
  run X=lib
  B=16 run X=lib
done
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
FLAGS="--precision native" run X=tip
FLAGS="--config 3" B=64 run X=tip
cat $O | cut -c1-400
