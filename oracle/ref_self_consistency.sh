#!/usr/bin/env bash
# TEST INFRASTRUCTURE.  How reproducible is the reference itself?  Runs oracle/_ref/ref_driver (the reference's own
# LibTorch CPU renderer) on the golden 8x8 scenes under different CPU dispatch settings and prints how far its OWN outputs
# move: the fine-pass depths are a discontinuous function of the coarse weights (searchsorted on CDF plateaus), so ulp-level
# changes in ATen's vector kernels / MKL's blocking move samples and pixels.  This is the floor for any parity claim.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
a="$(mktemp -d)"; b="$(mktemp -d)"; c="$(mktemp -d)"
trap 'rm -rf "$a" "$b" "$c"' EXIT
ATEN_CPU_CAPABILITY=avx2 "$here/_ref/ref_driver" golden "$a" >/dev/null 2>&1
ATEN_CPU_CAPABILITY=default "$here/_ref/ref_driver" golden "$b" >/dev/null 2>&1
MKL_CBWR=COMPATIBLE OMP_NUM_THREADS=1 "$here/_ref/ref_driver" golden "$c" >/dev/null 2>&1
python3 - "$a" "$b" "$c" <<'PY'
import sys, numpy as np
a, b, c = sys.argv[1:4]
for tag in ("render_hash", "render_classic", "render_hash_lindisp"):
    for other, name in ((b, "ATEN_CPU_CAPABILITY=default"), (c, "MKL_CBWR=COMPATIBLE, 1 thread")):
        L = lambda d, k: np.load(f"{d}/{tag}.{k}.npy")
        print(f"{tag:22s} avx2 vs {name:30s}: rgb max|diff| = {np.abs(L(a,'out_rgb')-L(other,'out_rgb')).max():.2e}   "
              f"fine depths identical = {(L(a,'fine_z')==L(other,'fine_z')).mean()*100:.1f} %")
PY
