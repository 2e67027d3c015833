#!/bin/bash
# Round 4, first GPU call: per-change A/B of the r4-prep chain from prebuilt libraries (profiles/scripts/prebuild_libs.sh), plus the two
# knobs the chain added (qr_prio, gemm_xcd) on its tip.   usage: bash profiles/scripts/r4_ab_chain.sh > gpurun_out/r4_ab_chain.txt
R=$GRAFT_REPO_ROOT
cd $R
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
export TRX_BENCH_NOPROF=1
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), 'layer-solves/s', round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'))
except Exception as e: print('FAILED', e)"; }
b128() { echo -n "  batch 128 $1: "; timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line; }
b16()  { echo -n "  batch 16  $1: "; timeout 200 python bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | line; }
gemm() { timeout 120 python tests/gpu_gemm_bench.py 2>&1 | grep -v amdgpu | grep "m= 1922 n= 1922 k= 1922 batch=128\|m=  961\|m= 4096" | sed "s/^/  $1 /"; }
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for lib in profiles/_ab_libs/*.so; do
  cp $lib torcwa_amd/libtrx.so
  n=$(basename $lib)
  echo "== $n"
  b128
  case $n in 00*|04*|06*) b16;; esac
  case $n in 00*|02*|05*) gemm;; esac
done
echo "== tip with knobs"
TRX_QR_PRIO=1 b128 qr_prio=1
TRX_QR_PRIO=1 b16 qr_prio=1
TRX_GEMM_XCD=1 b128 gemm_xcd=1
TRX_GEMM_XCD=1 gemm gemm_xcd=1
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
