#!/bin/bash
# Variants of the fp32 update arithmetic of eig_qr.hip (macro TRX_QR_VAR) as prebuilt libraries under profiles/_ab_libs/ (git-ignored; they travel
# with gpurun) for profiles/scripts/ab_prebuilt.sh:   bash profiles/scripts/build_qr_variants.sh 0 1 2 3 4
set -e
ROOT=$(git rev-parse --show-toplevel)
OUT=$ROOT/profiles/_ab_libs
mkdir -p $OUT
python $ROOT/torcwa_amd/csrc/build.py > /dev/null
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DTRX_QR_VAR=$v -c $ROOT/torcwa_amd/csrc/eig_qr.hip -o /tmp/eig_qr_var$v.o
  objs=$(ls $ROOT/torcwa_amd/csrc/_obj/*.o | grep -v eig_qr.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/1${v}_qrvar$v.so $objs /tmp/eig_qr_var$v.o
  echo "1${v}_qrvar$v.so"
done
