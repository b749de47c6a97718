#!/bin/bash
# helper run on the GPU box: tests + smoke + bench + rocprof kernel trace
set -u
tag=${1:-t}
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/${tag}_tests.log 2>&1
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > gpurun_out/${tag}_smoke.log 2>&1
(timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | tail -3) > gpurun_out/${tag}_bench_f16.log 2>&1
(timeout 600 python bench.py --steps 3 --warmup 1 --precision f32 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/${tag}_bench_f32.log 2>&1
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof -- python3 $ROOTD/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -5) > $ROOTD/gpurun_out/${tag}_prof.log 2>&1
cd $ROOTD
tail -4 gpurun_out/${tag}_tests.log
