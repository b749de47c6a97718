#!/bin/bash
# clock held under load + matrix-pipe busy cycles of the dominant kernels (separate --pmc passes, kernel dispatch only)
set -u
# counters are per dispatch: the Chunk loop on ONE stream, so that no two kernels run at the same time (the variable is inherited; nothing stands between rocprofv3's -- and python3)
export NRF_RENDER_LANES=1
tag=${1:-pmcclk}
ROOTD=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for wl in "hash f16x3" "hash f16" "classic f16x3" "classic f16"; do
  set -- $wl
  for grp in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES"; do
    name=$(echo ${1}_${2}_$grp | tr ' ' '_' | cut -c1-48)
    (timeout 600 rocprofv3 --pmc $grp --output-format csv -d $ROOTD/gpurun_out/${tag}_$name -- python3 $ROOTD/bench.py --workload $1 --precision $2 --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-parity --no-settle --no-isolated 2>&1 | tail -3) > $ROOTD/gpurun_out/${tag}_$name.log 2>&1
  done
done
cd $ROOTD
python3 - "$tag" <<'PY'
import csv, glob, sys, collections, json
tag = sys.argv[1]
out = {}
for d in sorted(glob.glob(f"gpurun_out/{tag}_*/")):
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("k_mlp_small_mfma", "k_mlp_nerf_mfma", "k_mlp_nerf_split", "k_sigma_small_f32", "k_sigma_nerf_f32", "k_hash_cu_lm")): continue
            short = "mlp_small" if "k_mlp_small" in k else ("mlp_nerf_split" if "k_mlp_nerf_split" in k else ("mlp_nerf" if "k_mlp_nerf" in k else ("sigma_nerf_f32" if "k_sigma_nerf" in k else ("sigma_small_f32" if "k_sigma" in k else "hash_encode"))))
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        run = d.rstrip("/").split(tag + "_")[1].split("_GRBM")[0].split("_SQ_")[0]
        for short, v in agg.items():
            for c, x in v.items():
                x = sorted(x)[len(x) // 4:]           # drop the short coarse-pass / tail dispatches: keep the large ones
                out.setdefault(run, {}).setdefault(short, {})[c] = sum(x) / len(x)
json.dump(out, open(f"gpurun_out/{tag}_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
rm -rf gpurun_out/${tag}_*/
