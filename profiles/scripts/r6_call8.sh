#!/bin/bash
# Round 6, eighth GPU call: where does the error of precision="native" come from (fused QR launches / sub-blocked LU panels on and off), GPU suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r6_call8.txt
: > $O
for env in "X=1" "TRX_QR_FUSE=1" "TRX_LU_SUB=1" "TRX_QR_FUSE=1 TRX_LU_SUB=1"; do
  env $env timeout 300 python tests/gpu_native_precision.py >> $O 2>>gpurun_out/r6_call8.err
done
echo "== gpu tests" >> $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 >> $O
cat $O | cut -c1-400
