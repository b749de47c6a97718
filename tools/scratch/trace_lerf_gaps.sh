R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_lerf -- python3 $R/tools/scratch/train_step_once.py lerf > /dev/null 2>&1
cd $R; f=$(ls gpurun_out/tr_lerf/*/*_kernel_trace.csv | head -1); python3 tools/scratch/trace_gaps.py $f 50 > gpurun_out/r6v_lerf_gaps.log 2>&1; rm -rf gpurun_out/tr_lerf; cat gpurun_out/r6v_lerf_gaps.log
