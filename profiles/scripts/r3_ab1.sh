#!/bin/bash
# Round 3, GPU call 1: A/B of the merged r3-prep prototypes (look-ahead schedule, banded unitary, LU row split, gemv geometry).
# usage (from the repo root on the GPU box): bash profiles/scripts/r3_ab1.sh > gpurun_out/r3_ab1.txt
export TRX_BENCH_NOPROF=1
run() { echo -n "$* : "; env "$@" timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strong-leg $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],1), d.get('numerical_failures'))"; }
echo "== batch 128"
EXTRA=""
run TRX_SLAB_BAND=1
run X=0
run TRX_QR_LOOK=2
run TRX_QR_LOOK=2 TRX_SLAB_WGS=320
run TRX_QR_LOOK=2 TRX_SLAB_WGS=512
run TRX_QR_LOOK=2 TRX_QR_GROUPS=2
run TRX_QR_LOOK=2 TRX_QR_GROUPS=8
run TRX_QR_LOOK=2 TRX_QR_AED=48
echo "== batch 16"
EXTRA="--batch 16"
run X=0
run TRX_QR_LOOK=2
run TRX_QR_LOOK=2 TRX_QR_GROUPS=1
run TRX_QR_LOOK=2 TRX_QR_GROUPS=4
run TRX_QR_LOOK=2 TRX_QR_AED=64
echo "== config 5 (n = 5202, forward + adjoint)"
python - <<'PY'
import sys, time, json, torch
sys.path.insert(0, '.')
import bench
from torcwa_amd.engine import Engine
args = bench.parse_args(['--config', '5'])
eng = Engine()
dev = torch.device('cuda')
rho = bench.make_inputs_topopt(dev)
def t(label, knobs):
    for k, v in knobs.items():
        eng.lib.check(eng.lib.tuning(k.encode(), v))
    ts = []
    for i in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        f = bench.run_step_topopt(rho, [25, 25], eng)
        torch.cuda.synchronize(); ts.append(time.time() - t0)
    print(label, ['%.2f' % x for x in ts], 'fom', complex(f.flatten()[0]).real, 'gnorm', bench._grad_norm[0], flush=True)
t('split default      ', {})
t('no LU split        ', {'lu_split': 1})
t('LU split again     ', {'lu_split': 0})
PY
