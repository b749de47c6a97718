#!/bin/bash
# rocprofv3 kernel summary of the LeRF training step (tools/scratch/lerf_train_time.py) -> gpurun_out/<tag>_lerf_train_kernel_stats.csv     usage: tools/scratch/lerf_train_prof.sh <tag>
tag=${1:-lt}
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
(timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof -- python3 $ROOTD/tools/scratch/lerf_train_time.py 2>&1 | tail -3) > $ROOTD/gpurun_out/${tag}_prof.log 2>&1
cd $ROOTD
f=$(ls gpurun_out/${tag}_prof/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_lerf_train_kernel_stats.csv; rm -rf gpurun_out/${tag}_prof
head -16 gpurun_out/${tag}_lerf_train_kernel_stats.csv | cut -c1-200
