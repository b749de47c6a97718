#!/bin/bash
# PMC passes focused on the NeRFSmall matrix-core kernel (one counter group per pass; kernel dispatch records only)
ROOTD=$PWD
prec=${1:-f16x3}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $ROOTD/gpurun_out/pmcmlp_${prec}_$i -- python3 $ROOTD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-also --precision $prec > /dev/null 2>&1
done
cd $ROOTD
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcmlp_${prec}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_mlp_small_mfma" in k or "k_hash_cu_lm" in k:
            acc["mlp" if "mlp" in k else "hash"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, d in acc.items():
    print(kn, {c: sum(v) / len(v) for c, v in sorted(d.items())})
PY
