#!/bin/bash
# Lanes per bulge of the chase kernel (profiles/scripts/build_lpb_variants.sh): GPU tests of the eigensolver + full-size golden cases with the 32-lane
# build, then 64 / 32 / 16 lanes side by side: default line, batch 16, all-fp64 route.
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5q_lpb.txt
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), d.get('parity_sample'))
except Exception as e: print('FAILED', e)"; }
cp profiles/_ab_libs/lpb32.so torcwa_amd/libtrx.so
echo "== GPU tests with lpb32.so" > $O
(timeout 170 python -m pytest tests/test_eig.py tests/test_fullsize_golden.py -m gpu -x -q 2>&1 | tail -3) >> $O
for v in 64 32 16; do
  cp profiles/_ab_libs/lpb$v.so torcwa_amd/libtrx.so
  echo "== lpb$v.so" >> $O
  echo -n "  batch 128      : " >> $O; timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line >> $O
  echo -n "  batch 16       : " >> $O; timeout 60 python bench.py --batch 16 --steps 4 --warmup 2 --no-cpu-baseline --no-strong-leg 2>/dev/null | line >> $O
  echo -n "  batch 128 fp64 : " >> $O; timeout 100 python bench.py --eig-route fp64 --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line >> $O
done
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
cat $O
