#!/bin/bash
# LDS counters of the NeRFSmall split kernels
R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in "NRF_SPLIT16=0" "NRF_SPLIT16_LDS=1"; do
  rm -rf /tmp/lp; env $v rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --output-format csv -d /tmp/lp -- python3 $R/tools/scratch/hash_time.py f16x3 > /tmp/lp.log 2>&1
  python3 - "$v" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/lp/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_mlp_small" in r["Kernel_Name"] and float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) > 1.0e6:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: "%.3e" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
done
tail -3 /tmp/lp.log
