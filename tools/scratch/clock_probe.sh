#!/bin/bash
# effective clock of the NeRFSmall kernel for tuning builds: GRBM_GUI_ACTIVE / 8 / duration
R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in main nolds samelds; do
  if [ $v = main ]; then unset NRF_LIB_PATH; else export NRF_LIB_PATH=$R/tune/$v/libnerfpp_hip.so; fi
  rm -rf /tmp/cp_$v; rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cp_$v -- python3 $R/tools/scratch/hash_time.py f16x3 > /tmp/cp_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
rows = []
for f in glob.glob(f"/tmp/cp_{v}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_mlp_small" in r["Kernel_Name"]:
            rows.append((float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
rows = [r for r in rows if r[1] > 1.0e6]          # the fine-pass dispatches (> 1 ms)
cyc = sum(r[0] for r in rows) / 8 / len(rows); ns = sum(r[1] for r in rows) / len(rows)
print(v, "dispatches", len(rows), "cycles/dispatch %.3e" % cyc, "ms %.3f" % (ns * 1e-6), "clock GHz %.3f" % (cyc / ns))
PY
done
