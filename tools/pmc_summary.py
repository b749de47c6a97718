#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc counter CSVs written by tools/gpu_pmc.sh into one JSON of per-kernel averages per dispatch.

usage: tools/pmc_summary.py gpurun_out/<tag> docs/history/profiles/round1/<name>.json
FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived metrics).  `hbm_bytes_per_launch` applies the gfx950 correction of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B fabric read requests at 64 B -> x2 on the read side."""
import collections
import csv
import glob
import json
import sys

tag, out = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(f"{tag}_*/runc/*_counter_collection.csv") + glob.glob(f"{tag}_*/*/*_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    launches = collections.Counter()
    instances = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = k.split("(")[0].replace("void ", "").strip()
        if "k_hash_cu_lm" in k: short = "hash_encode (k_hash_cu_lm)"
        elif "k_mlp_small_mfma" in k: short = "mlp_small (k_mlp_small_mfma)"
        elif "k_sigma_small_f32" in k: short = "sigma_small_f32 (k_sigma_small_f32)"
        elif "k_sigma_nerf_f32" in k: short = "sigma_nerf_f32 (k_sigma_nerf_f32)"
        elif "k_mlp_nerf_split" in k: short = "mlp_nerf_split (k_mlp_nerf_split)"
        elif "k_mlp_nerf_mfma" in k: short = "mlp_nerf (k_mlp_nerf_mfma)"
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        # one logical hash-encode launch is one dispatch PER INSTANCE of the kernel (round 5: the 12-levels-per-thread group + the 2-levels-per-thread group of the
        # four finest levels): dispatches / distinct instance names, settled below
        launches[(short, r["Counter_Name"])] += 1
        instances[short].add(k)
    for short, v in agg.items():
        for c, x in v.items():
            n = max(launches[(short, c)] // (len(instances[short]) if short.startswith("hash_encode") else 1), 1)
            res.setdefault(short, {})[c] = sum(x) / n
            res[short]["dispatches"] = len(x)
            res[short]["launches"] = n
for k, v in res.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
res = {k: v for k, v in res.items() if k.startswith(("hash_encode", "mlp_", "sigma_", "nrf::"))}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(f"wrote {out}: {len(res)} kernels")
