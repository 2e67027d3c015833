#!/bin/bash
# Lanes per bulge of the chase kernel (constant LPB of eig_qr.hip, patched in a temporary copy: 64 = one wave per bulge, 1024 threads; 32 = two bulges per
# wave; 16 = four)
# as prebuilt libraries under profiles/_ab_libs/ (git-ignored; they travel with gpurun):   bash profiles/scripts/build_lpb_variants.sh 64 32 16
set -e
ROOT=$(git rev-parse --show-toplevel)
OUT=$ROOT/profiles/_ab_libs
mkdir -p $OUT
python $ROOT/torcwa_amd/csrc/build.py > /dev/null
for v in "$@"; do
  sed "s/^constexpr int LPB = 64; /constexpr int LPB = $v; /" $ROOT/torcwa_amd/csrc/eig_qr.hip > /tmp/eig_qr_lpb$v.hip
  grep -q "^constexpr int LPB = $v; " /tmp/eig_qr_lpb$v.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I $ROOT/torcwa_amd/csrc -c /tmp/eig_qr_lpb$v.hip -o /tmp/eig_qr_lpb$v.o
  objs=$(ls $ROOT/torcwa_amd/csrc/_obj/*.o | grep -v eig_qr.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lpb$v.so $objs /tmp/eig_qr_lpb$v.o
  echo "lpb$v.so"
done
