#!/bin/bash
# A/B of a prototype library (torcwa_amd/libtrx_r3.so, built from branch r3-prep) on the box's copy of the repository:
# the off-window update with (default) and without (TRX_SLAB_BAND=1) the banded-unitary block skip.  usage: bash profiles/scripts/ab_band.sh
cp torcwa_amd/libtrx_r3.so torcwa_amd/libtrx.so
for v in 1 0; do
  echo -n "TRX_SLAB_BAND=$v : "
  TRX_SLAB_BAND=$v TRX_BENCH_NOPROF=1 timeout 70 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],1), d['parity_sample']['rel_err_vs_c128_oracle'] if 'parity_sample' in d else '')"
done
