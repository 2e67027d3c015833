#!/bin/bash
# One GPU call: every profiles/_ab_libs/*.so takes the place of torcwa_amd/libtrx.so and runs the bench at batch 128 and 16 (+ one accuracy figure)
R=$GRAFT_REPO_ROOT
cd $R
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), (d.get('parity_sample') or {}).get('txx00_rel_err_vs_c128_oracle'))
except Exception as e: print('FAILED', e)"; }
for lib in profiles/_ab_libs/*.so; do
  cp $lib torcwa_amd/libtrx.so
  echo "== $(basename $lib)"
  echo -n "  batch 128: "; timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
  echo -n "  batch 16 : "; timeout 200 python bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
done
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
