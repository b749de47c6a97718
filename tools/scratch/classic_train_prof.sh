#!/bin/bash
# rocprofv3 kernel summary of the classic training step (tools/scratch/classic_train_phases.py) -> gpurun_out/<tag>_classic_train_kernel_stats.csv
tag=${1:-ct}
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
(timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof -- python3 $ROOTD/tools/scratch/classic_train_phases.py 2>&1 | tail -3) > $ROOTD/gpurun_out/${tag}_prof.log 2>&1
cd $ROOTD
f=$(ls gpurun_out/${tag}_prof/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_classic_train_kernel_stats.csv; rm -rf gpurun_out/${tag}_prof
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_classic_train_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total ms %.1f (7 steps: per step %.1f)"%(tot/1e6, tot/7e6))
for r in rows[:14]:
    n=r["Name"]; n=(n[:60]+"..") if len(n)>62 else n
    print(n.ljust(64), r["Calls"].rjust(5), "%.2f ms/step"%(int(r["TotalDurationNs"])/7e6), "avg %.0f us"%(float(r["AverageNs"])/1e3))
PY
