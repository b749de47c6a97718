#!/bin/bash
# PMC passes for the dominant kernels (separate passes per counter group, no trace domains besides kernel dispatch)
set -u
tag=${1:-pmc}
ROOTD=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  (timeout 600 rocprofv3 --pmc $grp --output-format csv -d $ROOTD/gpurun_out/${tag}_$name -- python3 $ROOTD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-parity --no-settle --no-isolated ${BENCH_ARGS:-} 2>&1 | tail -3) > $ROOTD/gpurun_out/${tag}_$name.log 2>&1
done
cd $ROOTD
ls gpurun_out/${tag}_*/ 2>/dev/null | head
