#!/bin/bash
R=$PWD; mkdir -p $R/gpurun_out/trainprof; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trprof -o tr -- python3 $R/tools/scratch/train_prof.py 16384 f16 packed > $R/gpurun_out/trainprof/log.txt 2>&1
f=$(find /tmp/trprof -name "*kernel_stats.csv" | head -1)
head -32 "$f" | cut -c1-170 > $R/gpurun_out/trainprof/kernel_stats.txt
cat $R/gpurun_out/trainprof/kernel_stats.txt; grep -v "amdgpu.ids\|^W2026\|^E2026" $R/gpurun_out/trainprof/log.txt | tail -8
