#!/bin/bash
# HBM bytes of the training GEMM kernels (tools/scratch/gemm_one.py): FETCH_SIZE / WRITE_SIZE passes, per-kernel sums -> gpurun_out/<tag>_gemm_pmc.log
tag=${1:-g}; R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc_$grp -- python3 $R/tools/scratch/gemm_one.py > $R/gpurun_out/${tag}_pmc_$grp.log 2>&1
done
cd $R
python3 - <<'PY' > gpurun_out/${tag}_gemm_pmc.log
import csv, glob, collections, os, sys
tag = os.environ.get("TAG", "")
out = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/*_pmc_%s/*/*counter_collection.csv" % grp):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm" not in k and "tn_sum" not in k and "split_b" not in k: continue
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == grp: calls[(k, grp)] += 1
for k, v in out.items():
    n = max(1, calls[(k, "FETCH_SIZE")])
    # gfx950: FETCH_SIZE counts 128-byte fabric reads at 64 B -> x 2; both in KB
    print("%-60s calls %d  read %.3f GB  written %.3f GB per launch" % (k[:60], n, 2 * v["FETCH_SIZE"] * 1024 / n / 1e9, v["WRITE_SIZE"] * 1024 / max(1, calls[(k, "WRITE_SIZE")]) / 1e9))
PY
rm -rf gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE
cat gpurun_out/${tag}_gemm_pmc.log
