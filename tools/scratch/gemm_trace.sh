#!/bin/bash
# build (here): tools/scratch/gemm_trace.sh build    run (on the box): tools/scratch/gemm_trace.sh run   -- the -DNRF_GB_TRACE library under tune/trace, linked against the in-tree objects
R=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = build ]; then
  mkdir -p $R/tune/trace && cd $R/nerfpp_amd/csrc &&
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I../../include -I. -DNRF_GB_TRACE -c gemm_bf16x3.hip -o $R/tune/trace/gemm_bf16x3.o &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tune/trace/libnerfpp_hip.so $(ls ../lib/obj/*.o | grep -v gemm_bf16x3.o) $R/tune/trace/gemm_bf16x3.o && echo built
else
  NRF_LIB_PATH=$R/tune/trace/libnerfpp_hip.so timeout -k 10 200 python3 $R/tools/scratch/gemm_trace.py 2>&1 | grep -v amdgpu
fi
