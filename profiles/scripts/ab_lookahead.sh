#!/bin/bash
# A/B of the look-ahead schedule of the QR sweeps (needs the library built from the tree with profiles/prototypes/*.patch applied:
# knob TRX_QR_LOOK exists only there).  Default bench workload (batch 128), event timing off.  usage: bash profiles/scripts/ab_lookahead.sh
run() { echo -n "$* : "; env "$@" TRX_BENCH_NOPROF=1 timeout 90 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],1), d.get('numerical_failures'))"; }
run X=0
run TRX_QR_LOOK=2
run TRX_QR_LOOK=2 TRX_SLAB_WGS=384
run TRX_QR_LOOK=2 TRX_SLAB_WGS=512
run TRX_QR_LOOK=2 TRX_QR_GROUPS=2
run TRX_QR_LOOK=2 TRX_QR_GROUPS=2 TRX_SLAB_WGS=384
run TRX_QR_LOOK=2 TRX_QR_GROUPS=8
run TRX_SLAB_BAND=1
