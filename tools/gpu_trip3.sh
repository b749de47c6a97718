#!/bin/bash
# Round-3 helper run on the GPU box: tests + smoke + every bench line + the 2-rank rehearsal through bench.py's own launcher + rocprofv3 kernel-trace summaries
# (copied into profiles/round3 afterwards).   usage: tools/gpu_trip3.sh <tag>
set -u
tag=${1:-t}
mkdir -p gpurun_out
(timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/${tag}_tests.log 2>&1
(timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > gpurun_out/${tag}_smoke.log 2>&1
(timeout -k 10 900 python bench.py 2>gpurun_out/${tag}_bench_default.err | tail -1) > gpurun_out/${tag}_bench_default.json
(timeout -k 10 600 python bench.py --precision f16 --no-cpu-baseline --no-also 2>/dev/null | tail -1) > gpurun_out/${tag}_bench_hashnerf_f16.json
(timeout -k 10 600 python bench.py --steps 3 --warmup 1 --precision f32 --no-cpu-baseline --no-also 2>/dev/null | tail -1) > gpurun_out/${tag}_bench_hashnerf_f32.json
(timeout -k 10 600 python bench.py --workload classic --steps 5 --warmup 1 --no-also 2>/dev/null | tail -1) > gpurun_out/${tag}_bench_classic_f16x3.json
(timeout -k 10 600 python bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-parity 2>gpurun_out/${tag}_rehearsal.err | tail -1) > gpurun_out/${tag}_rehearsal_2ranks_one_gpu_gloo.json
(timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --backend gloo --steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-also 2>gpurun_out/${tag}_rehearsal_torchrun.err | tail -1) > gpurun_out/${tag}_rehearsal_2ranks_one_gpu_gloo_torchrun.json
(timeout -k 10 300 python bench.py --force-dist --collective cabi --scaling strong --no-also --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/${tag}_bench_strong_cabi_world1.json
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
(timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof -- python3 $ROOTD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-parity --no-isolated 2>&1 | tail -5) > $ROOTD/gpurun_out/${tag}_prof.log 2>&1
(export NRF_RENDER_LANES=1; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof_single_lane -- python3 $ROOTD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-parity 2>&1 | tail -5) > $ROOTD/gpurun_out/${tag}_prof_single_lane.log 2>&1
(timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof_classic -- python3 $ROOTD/bench.py --workload classic --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-parity 2>&1 | tail -5) > $ROOTD/gpurun_out/${tag}_prof_classic.log 2>&1
(timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof_lerf -- python3 $ROOTD/tools/scratch/lerf_time.py 2>&1 | tail -8) > $ROOTD/gpurun_out/${tag}_prof_lerf.log 2>&1
(timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/${tag}_prof_train -- python3 $ROOTD/tools/scratch/train_prof.py 16384 f16 binned --fast-only 2>&1 | tail -8) > $ROOTD/gpurun_out/${tag}_prof_train.log 2>&1
cd $ROOTD
for d in prof prof_single_lane prof_classic prof_lerf prof_train; do f=$(ls gpurun_out/${tag}_$d/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${d}_kernel_stats.csv; rm -rf gpurun_out/${tag}_$d; done
tail -4 gpurun_out/${tag}_tests.log
