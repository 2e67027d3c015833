#!/bin/bash
# Round 3, GPU call: where does the inverse-iteration route (eigenvalues-only QR + 7 TF-equivalent solve kernel) beat Schur vectors?
export TRX_BENCH_NOPROF=1
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],2), round(d['ms_per_step'],1), d.get('numerical_failures'))"; }
for b in 8 16 32 64 128; do
  EXTRA="--batch $b"
  echo "== batch $b"
  run TRX_EIG_VEC=1
  run TRX_EIG_VEC=2 TRX_INVIT_CFG=6
done
echo "== config 5 (n = 5202, batch 1)"
EXTRA="--config 5"
run TRX_EIG_VEC=1
run TRX_EIG_VEC=2 TRX_INVIT_CFG=4
run TRX_EIG_VEC=2 TRX_INVIT_CFG=0
