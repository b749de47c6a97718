#!/usr/bin/env python3
"""Per-kernel instruction histogram of one source file of nerfpp_amd/csrc, from the compiler's own assembly (no GPU needed).

usage: tools/isa_hist.py <file.hip> [kernel-name-substring] [-D...]     e.g.  tools/isa_hist.py mlp_nerf_split_mfma.hip k_mlp_nerf_split

For every kernel whose mangled name contains the substring: vector / matrix / scalar / LDS / vector-memory instruction counts of the code up to s_endpgm, the
scratch size, the VGPR count, and the most frequent opcodes.  What it was used for in round 3: gathers taken apart into dword loads
(v_cndmask / global_load_dword counts against the source), a dynamic vector-element extract compiled to seven v_cndmask per value, 64-bit per-lane address sums in
front of every LDS-DMA (v_lshl_add_u64), register-side offsets of hand-placed LDS reads (v_add_u32), by-value parameter structs living in scratch.
Static counts: loops are counted once, both sides of a branch are counted -- compare against the SQ_INSTS_* counters of a run (tools/gpu_pmc.sh) before concluding."""
import collections
import os
import re
import subprocess
import sys
import tempfile

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, "..", "nerfpp_amd", "csrc")
src = sys.argv[1]
key = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
extra = [a for a in sys.argv[2:] if a.startswith("-")]
if not os.path.exists(src):
    src = os.path.join(csrc, src)
stem = os.path.splitext(os.path.basename(src))[0]
# the per-file flags of the Makefile
flags = ["-fno-honor-nans"] if stem in ("mlp_small_mfma", "sigma_small_f32", "sigma_lerf_f32", "sigma_nerf_f32", "mlp_nerf_split_mfma", "mlp_lerf_split_mfma") else []
if stem == "mlp_small_bwd_mfma":
    flags = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, stem + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fvisibility=hidden", "-I" + os.path.join(here, "..", "include"),
           "-I" + csrc, "-S", "--cuda-device-only", "-o", out, src] + flags + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
lines = txt.split("\n")
meta = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    ps = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)); vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2))
    meta[m.group(1)] = (int(ps.group(1)) if ps else 0, int(vg.group(1)) if vg else 0)
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l]
for n, (i, name) in enumerate(starts):
    if name not in meta or key not in name:
        continue
    end = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    body = []
    for l in lines[i:end]:
        t = l.strip()
        if not l.startswith("\t") or t.startswith(".") or t.startswith(";"):
            continue
        body.append(t.split()[0])          # to the end of the function: a kernel with early exits has several s_endpgm
    c = collections.Counter(body)
    grp = lambda pred: sum(v for k, v in c.items() if pred(k))
    print(name)
    print("  total %d  valu %d  mfma %d  salu %d  lds %d  vmem %d  s_nop %d  scratch %d B  vgprs %d" % (
        len(body), grp(lambda k: k.startswith("v_") and not k.startswith("v_mfma")), grp(lambda k: k.startswith("v_mfma")), grp(lambda k: k.startswith("s_")),
        grp(lambda k: k.startswith("ds_")), grp(lambda k: k.startswith(("global_", "buffer_", "flat_", "scratch_"))), c.get("s_nop", 0), meta[name][0], meta[name][1]))
    print("  " + ", ".join("%s %d" % kv for kv in c.most_common(28)))
