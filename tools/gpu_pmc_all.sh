#!/bin/bash
# HBM-traffic / cache / SQ counter passes for both workloads, summarised on the box (the raw rocprofv3 output is too large to bring back)
set -u
# counters are per dispatch: the Chunk loop on ONE stream, so that no two kernels run at the same time (the variable is inherited; nothing stands between rocprofv3's -- and python3)
export NRF_RENDER_LANES=1
tag=${1:-pmc}
./tools/gpu_pmc.sh ${tag}h > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/${tag}h gpurun_out/${tag}_hashnerf.json
BENCH_ARGS="--workload classic" ./tools/gpu_pmc.sh ${tag}c > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/${tag}c gpurun_out/${tag}_classic.json
rm -rf gpurun_out/${tag}h_*/ gpurun_out/${tag}c_*/
ls -la gpurun_out/${tag}_*.json
