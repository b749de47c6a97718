mkdir -p gpurun_out
for nt in 8 16 32; do echo "threads $nt"; NRF_PACK_THREADS=$nt timeout -k 10 200 python tools/scratch/train_step_once.py lerf 2>/dev/null | grep '^{'; NRF_PACK_THREADS=$nt timeout -k 10 200 python tools/scratch/train_step_once.py classic 2>/dev/null | grep '^{'; done
