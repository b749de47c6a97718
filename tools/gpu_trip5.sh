#!/bin/bash
# Round-5 helper run on the GPU box: [tests +] smoke + the driver's own bench command (compact line on stdout, full record in bench_detail.json) and optional
# rocprofv3 kernel-trace summaries (copied into docs/history/profiles/round5 afterwards).   usage: tools/gpu_trip5.sh <tag> [tests] [prof]
set -u
tag=${1:-t}
shift
mkdir -p gpurun_out
want() { for a in "$@"; do :; done; }
has() { local k=$1; shift; for a in "$@"; do [ "$a" = "$k" ] && return 0; done; return 1; }
if has tests "$@"; then
(timeout -k 10 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -40) > gpurun_out/${tag}_tests.log 2>&1
(timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > gpurun_out/${tag}_smoke.log 2>&1
fi
(timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS:-} 2>gpurun_out/${tag}_bench_default.err | tail -1) > gpurun_out/${tag}_bench_default.json
cp bench_detail.json gpurun_out/${tag}_bench_detail.json 2>/dev/null
wc -c gpurun_out/${tag}_bench_default.json
if has prof "$@"; then
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
(timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof -- python3 $ROOTD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-parity --no-isolated 2>&1 | tail -5) > $ROOTD/gpurun_out/${tag}_prof.log 2>&1
(export NRF_RENDER_LANES=1; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof_single_lane -- python3 $ROOTD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-parity --no-isolated 2>&1 | tail -5) > $ROOTD/gpurun_out/${tag}_prof_single_lane.log 2>&1
cd $ROOTD
for d in prof prof_single_lane; do f=$(ls gpurun_out/${tag}_$d/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${d}_kernel_stats.csv; rm -rf gpurun_out/${tag}_$d; done
fi
[ -f gpurun_out/${tag}_tests.log ] && tail -4 gpurun_out/${tag}_tests.log
cat gpurun_out/${tag}_bench_default.json
