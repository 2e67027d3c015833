#!/bin/bash
# Round 6, third GPU call: increments of the QR changes from prebuilt libraries (round-5 tip, call-2 state, chase kernel, far streams),
# the failing eig tests with their full output, the GPU suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call3.txt
: > $O
cp torcwa_amd/libtrx.so /tmp/libtrx_tip.so
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph={p['phase'].split('/')[-1].strip(): round(p['ms_per_step']) for p in d['roofline']['phases']['inside_trx_eig']} if d.get('roofline') and d['roofline'].get('phases') else {}
    print(round(d['value'],4), d['unit'], round(d['ms_per_step'],1), 'ms', d.get('numerical_failures'), ph)
except Exception as e: print('FAILED', e)"; }
run() { echo -n "$* B=${B:-128} ${FLAGS}: " >> $O; env "$@" timeout 400 python bench.py --batch ${B:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg ${FLAGS} 2>>gpurun_out/r6_call3.err | line >> $O; }
echo "== increments (prebuilt libraries; phases inside trx_eig in ms per step)" >> $O
for lib in profiles/_ab_libs/*.so; do
  cp $lib torcwa_amd/libtrx.so
  echo "-- $(basename $lib)" >> $O
  run X=lib
  B=16 run X=lib
done
cp /tmp/libtrx_tip.so torcwa_amd/libtrx.so
echo "-- tip (far streams)" >> $O
run X=tip
run TRX_QR_FAR=1
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=8 TRX_QR_GROUPS=3
run GPU_MAX_HW_QUEUES=8 TRX_QR_SUPER=8
run TRX_QR_SUPER=2
B=16 run X=tip
B=16 run TRX_QR_FAR=1
B=16 run GPU_MAX_HW_QUEUES=8
B=16 run GPU_MAX_HW_QUEUES=8 TRX_QR_GROUPS=4
B=64 run X=tip
B=32 run X=tip
echo "== QR cycle counters (batch 16, tip)" >> $O
TRX_QR_DEBUG=1 timeout 300 python bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "libtrx qr" | tail -2 >> $O
echo "== test_eig on the GPU, full output of failures" >> $O
timeout 900 python -m pytest tests/test_eig.py -m gpu -q 2>&1 | tail -120 >> $O
echo "== gpu tests" >> $O
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_eig.py 2>&1 | tail -25 >> $O
cat $O | cut -c1-400
