#!/usr/bin/env python3
"""Merge the per-kernel PMC summaries of one round into profiles/pmc_latest.json (what bench.py reads for roofline.traffic / mfma_busy).

usage: tools/pmc_merge.py <hashnerf.json> <classic.json> <clock_summary.json> <commit> <out.json>
  * HBM bytes per point = (2 * FETCH_SIZE + WRITE_SIZE) KB per logical launch (gfx950: FETCH_SIZE counts 128-B fabric reads at 64 B) / points per launch of
    `bench.py --steps 1 --warmup 1 --no-also --no-parity [--workload classic]` (2 frames: 640 000 rays each; executed points per ray from bench.executed_per_ray);
  * matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, separate passes; clock = GRBM_GUI_ACTIVE / 8 / kernel duration is not
    available from counters alone and is left to tools/scratch/clock_probe.sh;
  * _meta.commit and _meta.kernel_source_sha256_16: bench.py drops `traffic` when a kernel's sources no longer match."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import costs as bench  # noqa: E402  (constants; touches no GPU)
from benchlib import roofline as _rf  # noqa: E402
bench.kernel_source_hash = _rf.kernel_source_hash
bench.PMC_KERNEL_SOURCES = _rf.PMC_KERNEL_SOURCES

hn, cl, clk, commit, out = sys.argv[1:6]
res = {}
for f in (hn, cl):
    if os.path.exists(f):
        res.update(json.load(open(f)))
FRAMES = 2                      # --steps 1 --warmup 1
RAYS = bench.H * bench.W
per_ray = {"hash_encode (k_hash_cu_lm)": bench.executed_per_ray("hash", "f16x3", "cu")[0],
           "mlp_small (k_mlp_small_mfma)": bench.executed_per_ray("hash", "f16x3", "cu")[1] + bench.colour_only_per_ray("hash", "f16x3"),      # both instances (whole network / colour net alone) are summed under this name
           "sigma_small_f32 (k_sigma_small_f32)": bench.executed_per_ray("hash", "f16x3", "cu")[2],
           "mlp_nerf_split (k_mlp_nerf_split)": bench.executed_per_ray("classic", "f16x3", "cu")[1]}
points = {}
for k, v in res.items():
    if k in per_ray and "hbm_bytes_per_launch" in v:
        pts = FRAMES * RAYS * per_ray[k]
        points[k] = pts
        v["hbm_bytes_per_point"] = v["hbm_bytes_per_launch"] * v["launches"] / pts
        v["points_in_run"] = pts
if os.path.exists(clk):
    c = json.load(open(clk))
    names = {"mlp_small": "mlp_small (k_mlp_small_mfma)", "mlp_nerf_split": "mlp_nerf_split (k_mlp_nerf_split)", "mlp_nerf": "mlp_nerf (k_mlp_nerf_mfma)",
             "sigma_small_f32": "sigma_small_f32 (k_sigma_small_f32)", "sigma_nerf_f32": "sigma_nerf_f32 (k_sigma_nerf_f32)", "hash_encode": "hash_encode (k_hash_cu_lm)"}
    for run, kernels in c.items():            # run = "<workload>_<precision>"
        prec = run.split("_")[-1]
        for short, ctr in kernels.items():
            key = names.get(short)
            if not key:
                continue
            e = res.setdefault(key, {})
            if "SQ_VALU_MFMA_BUSY_CYCLES" in ctr and "GRBM_GUI_ACTIVE" in ctr and ctr["GRBM_GUI_ACTIVE"] > 0:
                e.setdefault("mfma_busy_frac_of_active_cycles", {})[prec] = (ctr["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (ctr["GRBM_GUI_ACTIVE"] / 8.0)
            if "SQ_VALU_MFMA_COEXEC_CYCLES" in ctr and "GRBM_GUI_ACTIVE" in ctr and ctr["GRBM_GUI_ACTIVE"] > 0:       # same normalisation as the busy share: per SIMD over per XCD
                e.setdefault("mfma_valu_coexec_frac_of_active_cycles", {})[prec] = (ctr["SQ_VALU_MFMA_COEXEC_CYCLES"] / 1024.0) / (ctr["GRBM_GUI_ACTIVE"] / 8.0)
res["_meta"] = {
    "commit": commit,
    "kernel_source_sha256_16": {k: bench.kernel_source_hash(k) for k in bench.PMC_KERNEL_SOURCES},
    "points_in_run": points,
    "hbm_bytes": "(2*FETCH_SIZE + WRITE_SIZE) KB per dispatch (gfx950: FETCH_SIZE counts 128-B fabric reads at 64 B), summed over the run and divided by the points the kernel processed in it",
    "source": "rocprofv3 --pmc passes (one counter group per pass: FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS | TCP->TCC | SQ_*; GRBM_GUI_ACTIVE | SQ_VALU_MFMA_BUSY_CYCLES ... | "
              "SQ_VALU_MFMA_COEXEC_CYCLES ...) of `bench.py --steps 1 --warmup 1 --no-also --no-parity [--workload classic]`, tools/gpu_pmc_round.sh at commit " + commit,
}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, "kernels:", [k for k in res if k != "_meta"])
