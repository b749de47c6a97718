R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hp -- python3 $R/tools/scratch/train_prof.py 16384 f16 binned --fast-only > $R/gpurun_out/r8b_hash_train_prof.log 2>&1
cd $R; f=$(ls gpurun_out/hp/*/*_kernel_stats.csv | head -1); cp $f gpurun_out/r8b_hash_train_kernel_stats.csv; t=$(ls gpurun_out/hp/*/*_kernel_trace.csv | head -1); python3 tools/scratch/trace_gaps.py $t 30 > gpurun_out/r8b_hash_train_gaps.log 2>&1; rm -rf gpurun_out/hp
grep -v amdgpu gpurun_out/r8b_hash_train_prof.log | grep -v rocprofv3 | tail -8; head -30 gpurun_out/r8b_hash_train_gaps.log
