#!/usr/bin/env python3
"""Lint: does any inline-asm instruction of nerfpp_amd/csrc read a register that a matrix, packed-fp32 or transcendental instruction wrote last?

The compiler pads the hazards of its own instructions; it does not look into inline asm.  An asm consumer of such a result is right alone and wrong when other waves
load the pipe the producer runs in (round 3: the baked hash lookup with packed weight multiplies).  This script compiles every .hip of csrc to
assembly, finds the instructions between ;;#ASMSTART / ;;#ASMEND, and reports those whose vector sources were last written -- in straight-line order, a heuristic -- by
v_mfma_*, v_pk_*_f32 or a transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos).  usage: tools/asm_input_lint.py [file.hip ...]   exit code 1 on findings."""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, "..", "nerfpp_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(csrc, "*.hip")))
NOHONOR = ("mlp_small_mfma", "sigma_small_f32", "sigma_lerf_f32", "sigma_nerf_f32", "mlp_nerf_split_mfma", "mlp_lerf_split_mfma")
RISKY = re.compile(r"^(v_mfma_|v_smfmac_|v_pk_(add|mul|fma|mov)_f32|v_(rcp|rsq|sqrt|exp|log|sin|cos)_)")


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return [(tok[0], k) for k in range(int(m.group(1)), int(m.group(2)) + 1)]
    m = re.match(r"([va])(\d+)$", tok)
    return [(m.group(1), int(m.group(2)))] if m else []


def lint(src):
    if not os.path.exists(src):
        src = os.path.join(csrc, src)
    stem = os.path.splitext(os.path.basename(src))[0]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, stem + ".s")
        cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fvisibility=hidden", "-I" + os.path.join(here, "..", "include"),
               "-I" + csrc, "-S", "--cuda-device-only", "-o", out, src] + (["-fno-honor-nans"] if stem in NOHONOR else [])
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    findings, kernel, writer, in_asm, n_asm = [], "", {}, False, 0
    for l in lines:
        t = l.strip()
        if l.startswith("_Z") and ":" in l:
            kernel, writer = l.split(":")[0], {}
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l.startswith("\t") or t.startswith(";") or t.startswith("."):
            continue
        parts = t.split(None, 1)
        op, ops = parts[0], (parts[1].split(",") if len(parts) > 1 else [])
        if in_asm and op.startswith("v_"):
            n_asm += 1
            for srcop in ops[1:]:
                for r in regs(srcop.split()[0] if srcop.strip() else ""):
                    w = writer.get(r)
                    if w and RISKY.match(w):
                        findings.append((kernel, op, "%s%d" % r, w))
        if ops and (op.startswith(("v_", "global_load", "buffer_load", "ds_read", "scratch_load", "flat_load"))):
            for r in regs(ops[0].split()[0]):
                writer[r] = op
            if op.startswith("v_mfma") or op.startswith("v_pk_"):          # a second destination register range is not parsed separately: the first operand is the destination
                pass
    return stem, n_asm, findings


with ThreadPoolExecutor(max_workers=6) as ex:
    results = list(ex.map(lint, files))
bad = 0
for stem, n_asm, findings in results:
    print("%-24s %5d asm vector instructions, %d fed by a matrix / packed-fp32 / transcendental result" % (stem, n_asm, len(findings)))
    for k, op, r, w in findings[:10]:
        print("    %s: %s reads %s last written by %s" % (k[:70], op, r, w))
    bad += len(findings)
sys.exit(1 if bad else 0)
