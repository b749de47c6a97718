#!/bin/bash
# every fuzzer of tools/scratch with the seed given (a new one per round), logs under gpurun_out/<tag>_fuzz_*.log.  usage: fuzz_all.sh <tag> <seed>
tag=${1:-fz}; seed=${2:-1}
mkdir -p gpurun_out
rc=0
for f in render_fuzz:40 lane_fuzz:30 oracle_fuzz:30 lerf_fuzz:20 stage_fuzz:10 train_fuzz:16 hash_lm_fuzz:30 arch_fuzz:20; do
  name=${f%%:*}; n=${f#*:}
  timeout -k 10 420 python tools/scratch/$name.py $n $seed > gpurun_out/${tag}_fuzz_$name.log 2>&1; r=$?
  echo "$name rc=$r: $(grep -v amdgpu.ids gpurun_out/${tag}_fuzz_$name.log | tail -1 | cut -c1-300)"
  [ $r -ne 0 ] && rc=1
done
timeout -k 10 300 python tools/scratch/train_ops_fuzz.py 8 > gpurun_out/${tag}_fuzz_train_ops_fuzz.log 2>&1; echo "train_ops_fuzz rc=$?: $(tail -1 gpurun_out/${tag}_fuzz_train_ops_fuzz.log | cut -c1-300)"
timeout -k 10 300 python tools/scratch/concurrency_fuzz.py 8 > gpurun_out/${tag}_fuzz_concurrency_fuzz.log 2>&1; echo "concurrency_fuzz rc=$?: $(tail -1 gpurun_out/${tag}_fuzz_concurrency_fuzz.log | cut -c1-300)"
exit $rc
