#!/bin/bash
# One GPU call, several library builds: every profiles/_ab_libs/*.so in turn takes the place of torcwa_amd/libtrx.so (in the scratch copy of the
# repository on the GPU box) and runs the default bench, batch 16 and the hot GEMM shape.   usage: bash profiles/scripts/ab_prebuilt.sh
R=$GRAFT_REPO_ROOT
cd $R
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
for lib in profiles/_ab_libs/*.so; do
  cp $lib torcwa_amd/libtrx.so
  echo "== $(basename $lib)"
  echo -n "  batch 128: "; timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
  echo -n "  batch 16 : "; timeout 200 python bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line
  timeout 120 python tests/gpu_gemm_bench.py 2>&1 | grep -v amdgpu | grep "m= 1922 n= 1922 k= 1922 batch=128\|m=  961" | sed 's/^/  /'
done
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
