#!/bin/bash
# Timing-only builds of the whole-row GEMM kernel with parts left out (-DNRF_GB_ABL=<bits>, gemm_bf16x3.hip) -> tune/abl_<bits>/libnerfpp_hip.so, linked against
# the objects of the in-tree build.   usage (here, no GPU): tools/scratch/gemm_ablate.sh build 1 2 4 ...     (on the box): tools/scratch/gemm_ablate.sh run 0 1 2 4 ...
set -e
mode=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$mode" = build ]; then
  cd $R/nerfpp_amd/csrc
  for v in "$@"; do
    mkdir -p $R/tune/abl_$v
    /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I../../include -I. -DNRF_GB_ABL=$v -c gemm_bf16x3.hip -o $R/tune/abl_$v/gemm_bf16x3.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tune/abl_$v/libnerfpp_hip.so $(ls ../lib/obj/*.o | grep -v gemm_bf16x3.o) $R/tune/abl_$v/gemm_bf16x3.o
    echo built $v
  done
else
  for v in "$@"; do
    if [ "$v" = 0 ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$R/tune/abl_$v/libnerfpp_hip.so; fi
    echo "abl $v: $(timeout -k 10 120 python3 $R/tools/scratch/gemm_time.py 786432 256 256 2>&1 | grep -v amdgpu | tail -1)"
  done
fi
