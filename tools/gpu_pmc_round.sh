#!/bin/bash
# Refresh profiles/pmc_latest.json from the CURRENT kernels: HBM-traffic / cache / SQ counter passes (tools/gpu_pmc_all.sh) and the matrix-pipe / clock passes
# (tools/gpu_pmc_clock.sh), merged on the box by tools/pmc_merge.py, which stamps the summary with the commit and a hash of each kernel's sources --
# bench.py reports roofline.traffic only while those hashes still match.   usage (on the GPU box): tools/gpu_pmc_round.sh <tag> <commit>
set -u
tag=${1:-pmc}
commit=${2:-unknown}
./tools/gpu_pmc_all.sh ${tag} > gpurun_out/${tag}_all.log 2>&1
./tools/gpu_pmc_clock.sh ${tag}clk > gpurun_out/${tag}_clock.log 2>&1
python3 tools/pmc_merge.py gpurun_out/${tag}_hashnerf.json gpurun_out/${tag}_classic.json gpurun_out/${tag}clk_summary.json "$commit" gpurun_out/${tag}_pmc_latest.json
ls -la gpurun_out/${tag}_*.json
