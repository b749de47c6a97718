#!/bin/bash
# Builds the threads-as-ranks harness of the C-ABI collective (tests/helpers/comm_ranks_as_threads.cpp) over the mock RCCL (tests/helpers/mock_rccl.cpp, under RCCL's
# SONAME so that comm.hip's dlopen finds it) into tests/helpers/_build/.  Needs hipcc; run by tests/test_gpu_parity.py and by hand.   usage: tests/helpers/build_mock_rccl.sh
set -e
here="$(cd "$(dirname "$0")" && pwd)"; root="$here/../.."; out="$here/_build"
mkdir -p "$out"
/opt/rocm/bin/hipcc -std=c++17 -O2 -fPIC -shared --offload-arch=gfx950 -Wl,-soname,librccl.so.1 -I/opt/rocm/include "$here/mock_rccl.cpp" -o "$out/librccl.so.1"
/opt/rocm/bin/hipcc -std=c++17 -O2 -I"$root/include" "$here/comm_ranks_as_threads.cpp" -o "$out/comm_ranks_as_threads" -L"$out" -Wl,--no-as-needed -l:librccl.so.1 -Wl,--as-needed -L"$root/nerfpp_amd/lib" -lnerfpp_hip \
    -Wl,--disable-new-dtags -Wl,-rpath,"$out" -Wl,-rpath,"$root/nerfpp_amd/lib" -lpthread        # RPATH, not RUNPATH: searched BEFORE LD_LIBRARY_PATH, where the real librccl.so.1 lives
echo "built $out/comm_ranks_as_threads"
