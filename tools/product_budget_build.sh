#!/bin/bash
# Builds one libnerfpp_hip.so per product-budget mask of the split NeRFSmall kernel (NRF_SMALL_DROP_MASK, mlp_small_mfma.hip) into tune/pb_<mask>/: only that file is
# recompiled, the other objects come from nerfpp_amd/lib/obj.  usage: tools/product_budget_build.sh <mask> [<mask> ...]      (hex masks, e.g. 0x0 0x4 0x8)
set -e
cd "$(dirname "$0")/.."
CS=nerfpp_amd/csrc
FLAGS="-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Iinclude -I$CS -Wall -Wno-unused-function -fno-honor-nans"
others=$(ls nerfpp_amd/lib/obj/*.o | grep -v mlp_small_mfma.o)
build() {
  m=$1; d=tune/pb_$m; mkdir -p $d
  /opt/rocm/bin/hipcc $FLAGS -DNRF_SMALL_DROP_MASK=$m -c $CS/mlp_small_mfma.hip -o $d/mlp_small_mfma.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/libnerfpp_hip.so $d/mlp_small_mfma.o $others
  rm -f $d/mlp_small_mfma.o
  echo built $d
}
n=0
for m in "$@"; do build $m & n=$((n+1)); if [ $((n % 4)) -eq 0 ]; then wait; fi; done
wait
