#!/bin/bash
# kernel-level timing of the fp32 and matrix-core NeRFSmall backward
R=$PWD; mkdir -p $R/gpurun_out/bwd; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bwdprof -o bwd -- python3 $R/tools/scratch/bwd_time.py > $R/gpurun_out/bwd/log.txt 2>&1
f=$(find /tmp/bwdprof -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-200 > $R/gpurun_out/bwd/kernel_stats.txt
cat $R/gpurun_out/bwd/kernel_stats.txt; grep -v amdgpu.ids $R/gpurun_out/bwd/log.txt | tail -3
