#!/bin/bash
# rocprofv3 kernel summaries of the classic and the LeRF training step (tools/scratch/train_step_once.py) -> gpurun_out/<tag>_{classic,lerf}_kernel_stats.csv
tag=${1:-t}
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in classic lerf; do
  (timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_$w -- python3 $R/tools/scratch/train_step_once.py $w 2>&1 | grep -E '^\{' ) > $R/gpurun_out/${tag}_${w}_step.log 2>&1
  f=$(ls $R/gpurun_out/${tag}_prof_$w/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/${tag}_${w}_kernel_stats.csv
  rm -rf $R/gpurun_out/${tag}_prof_$w
done
cd $R
cat gpurun_out/${tag}_classic_step.log gpurun_out/${tag}_lerf_step.log
