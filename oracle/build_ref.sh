#!/usr/bin/env bash
# build_ref.sh -- TEST INFRASTRUCTURE. Compiles the reference's own LibTorch CPU path
# (from the sources where they lie under /root/reference/src) together with
# oracle/ref/ref_driver.cpp into oracle/_ref/ref_driver.
#
#   * No reference source is copied into this repository; outputs go only to oracle/_ref/
#     (git-ignored; travels to the GPU box like the repo's own built .so files).
#   * No stand-in headers or libraries are written.  Two reference files cannot be fed to
#     g++ on Linux verbatim, so they are FILTERED ON THE FLY into a mktemp dir that is
#     deleted at exit:
#       - NeRF.cpp:236  `torch::tensor({(1ll << ...) - 1ll}, kLong)` is ambiguous where
#         int64_t is `long` (the reference is MSVC-flavoured); the literal gets an
#         int64_t cast.  Arithmetic unchanged.
#       - NeRFRenderer.h:7-9,46-68  three <opencv2/...> includes and the two cv::Mat <->
#         Tensor image helpers (CVMatToTorchTensor / TorchTensorToCVMat) are deleted.
#         They are image I/O for the caller (NeRFExecutor::RenderPath), not on the
#         render path; OpenCV is absent from this image.  Every line of
#         Render/BatchifyRays/RenderRays/RunNetwork/RawToOutputs compiles unmodified.
#       - LeRFRenderer.h goes through the same temp dir UNCHANGED, only so that its
#         #include "NeRFRenderer.h" resolves to the filtered header.
#   * CUDA-only units (CuHashEmbedder.cu, CuSHEncoder.cu) and units that need RuCLIP /
#     COLMAP / OpenCV proper (NeRFExecutor.h, loaders) are NOT built: unbuildable here
#     (see DESIGN.md).  LeRFRenderer.cpp is compiled for adapter_lerf_check ONLY (not for
#     the oracle / ref_driver), with its RuCLIP include filtered out: see below.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
REF="${NRF_REFERENCE_DIR:-/root/reference}/src"
if [ ! -d "$REF" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) -- keeping prebuilt oracle/_ref if any" >&2
  exit 0
fi
T="$(python3 -c 'import torch,os;print(os.path.dirname(torch.__file__))')"
out="$here/_ref"
mkdir -p "$out"
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT

# filtered header (see above); everything else is included straight from $REF
sed -e '/^#include <opencv2\//d' \
    -e '/^inline torch::Tensor CVMatToTorchTensor/,/^}/d' \
    -e '/^inline cv::Mat TorchTensorToCVMat/,/^}/d' \
    "$REF/NeRFRenderer.h" > "$tmp/NeRFRenderer.h"
if grep -q 'cv::' "$tmp/NeRFRenderer.h"; then echo "filter failed: cv:: still referenced" >&2; exit 1; fi
# LeRFRenderer.h says #include "NeRFRenderer.h", which a compiler resolves next to the including file first: passed through the same temp dir (unchanged),
# it picks up the filtered header above instead of the OpenCV-including one beside it
cat "$REF/LeRFRenderer.h" > "$tmp/LeRFRenderer.h"

CXX="${CXX:-g++}"
FLAGS="-std=c++17 -O2 -fPIC -D_GLIBCXX_USE_CXX11_ABI=1 -D__HIP_PLATFORM_AMD__ -DUSE_ROCM -w"
INC="-I$tmp -I$REF -I$REF/LibTorchTraining -I$REF/Common -I$here/../include -I$T/include -I$T/include/torch/csrc/api/include -I/opt/rocm/include"
LIBS="-L$T/lib -Wl,-rpath,$T/lib -ltorch -ltorch_cpu -lc10"

# objects are cached in oracle/_ref/obj (object code only); rebuilt when the source is newer
mkdir -p "$out/obj"
stale() { [ ! -f "$1" ] || [ "$2" -nt "$1" ] || [ "$0" -nt "$1" ]; }
if stale "$out/obj/NeRF.o" "$REF/NeRF.cpp"; then
  sed -e 's/torch::tensor({(1ll << static_cast<long long>(log2_hashmap_size)) - 1ll}, torch::kLong)/torch::tensor({static_cast<int64_t>((1ll << static_cast<long long>(log2_hashmap_size)) - 1ll)}, torch::kLong)/' \
      "$REF/NeRF.cpp" | $CXX $FLAGS $INC -x c++ -c - -o "$out/obj/NeRF.o" &
fi
if stale "$out/obj/CustomOps.o" "$REF/CustomOps.cpp"; then $CXX $FLAGS $INC -c "$REF/CustomOps.cpp" -o "$out/obj/CustomOps.o" & fi
if stale "$out/obj/LeRF.o" "$REF/LeRF.cpp"; then $CXX $FLAGS $INC -c "$REF/LeRF.cpp" -o "$out/obj/LeRF.o" & fi
if stale "$out/obj/ref_driver.o" "$here/ref/ref_driver.cpp"; then $CXX $FLAGS $INC -c "$here/ref/ref_driver.cpp" -o "$out/obj/ref_driver.o" & fi
wait
$CXX -o "$out/ref_driver" "$out/obj/ref_driver.o" "$out/obj/NeRF.o" "$out/obj/CustomOps.o" "$out/obj/LeRF.o" $LIBS
echo "built $out/ref_driver"

# adapter_check: the LibTorch adapter (include/nerfpp_torch.h) inside the reference's NeRFRenderer<> machinery, against the
# reference CPU renderer.  Links the repo's own libnerfpp_hip.so (rpath relative to the binary) and LibTorch's HIP backend.
hiplib="$here/../nerfpp_amd/lib/libnerfpp_hip.so"
if [ -f "$hiplib" ]; then
  # adapter_check.cpp (the checks) and adapter_bench.cpp (`adapter_check bench ...`: the drop-in on the host's clock) share adapter_util.h; compiled side by side
  for unit in adapter_check adapter_bench; do
    if stale "$out/obj/$unit.o" "$here/ref/$unit.cpp" || [ "$here/ref/adapter_util.h" -nt "$out/obj/$unit.o" ] || [ "$here/../include/nerfpp_torch.h" -nt "$out/obj/$unit.o" ] || [ "$here/../include/nerfpp_hip.h" -nt "$out/obj/$unit.o" ]; then
      rm -f "$out/obj/$unit.o"
      $CXX $FLAGS $INC -I"$here/ref" -c "$here/ref/$unit.cpp" -o "$out/obj/$unit.o" &
    fi
  done
  wait
  for unit in adapter_check adapter_bench; do [ -f "$out/obj/$unit.o" ] || { echo "build_ref.sh: $unit.cpp did not compile" >&2; exit 1; }; done
  # HipLeRFRenderer : LeRFRenderer LINKED AND RUN (adapter_lerf_check): the reference's LeRFRenderer.cpp compiled from where it lies with its line 2 -- the include of the
  # external RuCLIP module's header, absent from the reference tree -- filtered out on the fly; the one symbol that unit takes from it (`Relevancy`, :79) is declared by a
  # forced include and DEFINED in adapter_lerf_check.cpp from this repository's own restatement.  Test infrastructure for the SUBCLASS only: it pins nothing (the file says so).
  if stale "$out/obj/LeRFRenderer.o" "$REF/LeRFRenderer.cpp"; then
    echo 'torch::Tensor Relevancy(torch::Tensor embeds, torch::Tensor positives, torch::Tensor negatives);' > "$tmp/relevancy_decl.h"
    sed -e '/^#include "RuCLIPProcessor.h"/d' "$REF/LeRFRenderer.cpp" | $CXX $FLAGS $INC -include torch/torch.h -include "$tmp/relevancy_decl.h" -x c++ -c - -o "$out/obj/LeRFRenderer.o" &
  fi
  if stale "$out/obj/adapter_lerf_check.o" "$here/ref/adapter_lerf_check.cpp" || [ "$here/ref/adapter_util.h" -nt "$out/obj/adapter_lerf_check.o" ] || [ "$here/../include/nerfpp_torch.h" -nt "$out/obj/adapter_lerf_check.o" ] || [ "$here/../include/nerfpp_hip.h" -nt "$out/obj/adapter_lerf_check.o" ]; then
    rm -f "$out/obj/adapter_lerf_check.o"
    $CXX $FLAGS $INC -I"$here/ref" -c "$here/ref/adapter_lerf_check.cpp" -o "$out/obj/adapter_lerf_check.o" &
  fi
  wait
  [ -f "$out/obj/adapter_lerf_check.o" ] && [ -f "$out/obj/LeRFRenderer.o" ] || { echo "build_ref.sh: adapter_lerf_check did not compile" >&2; exit 1; }
  $CXX -o "$out/adapter_lerf_check" "$out/obj/adapter_lerf_check.o" "$out/obj/LeRFRenderer.o" "$out/obj/NeRF.o" "$out/obj/CustomOps.o" "$out/obj/LeRF.o" $LIBS \
      -Wl,--no-as-needed -ltorch_hip -lc10_hip -Wl,--as-needed -L"$here/../nerfpp_amd/lib" -lnerfpp_hip \
      -Wl,-rpath,'$ORIGIN/../../nerfpp_amd/lib' -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64
  echo "built $out/adapter_lerf_check"
  $CXX -o "$out/adapter_check" "$out/obj/adapter_check.o" "$out/obj/adapter_bench.o" "$out/obj/NeRF.o" "$out/obj/CustomOps.o" "$out/obj/LeRF.o" $LIBS \
      -Wl,--no-as-needed -ltorch_hip -lc10_hip -Wl,--as-needed -L"$here/../nerfpp_amd/lib" -lnerfpp_hip \
      -Wl,-rpath,'$ORIGIN/../../nerfpp_amd/lib' -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64
  echo "built $out/adapter_check"
fi
