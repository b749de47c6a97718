R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_lerf -- python3 $R/tools/scratch/train_step_once.py ${1:-lerf} > /dev/null 2>&1
cd $R; f=$(ls gpurun_out/tr_lerf/*/*_kernel_trace.csv | head -1); python3 tools/scratch/trace_last_step.py $f > gpurun_out/r8i_${1:-lerf}_last_step.log 2>&1; rm -rf gpurun_out/tr_lerf; cat gpurun_out/r8i_${1:-lerf}_last_step.log
