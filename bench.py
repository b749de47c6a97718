#!/usr/bin/env python3
"""bench.py -- ray-samples/s of the HIP volume-rendering path on synthetic Blender-Lego-shaped frames.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU.  Either launcher works: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks
read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; a plain `python bench.py --gpus N` starts the N ranks itself (the parent
touches no GPU API, starts N fresh child processes, relays rank 0's single JSON line and exits non-zero if any child fails).

A "step" at N = 1 is one 800x800 frame (64 + 128 samples per ray, 256 network evaluations per ray).  At N > 1 the headline is BASELINE config 4,
strong scaling: a step is still ONE frame, cut into N contiguous row tiles; rank r renders tile r with one library call (nrf_render_rows) and one RCCL
all-gather hands every rank the complete frame -- `value` is the frame's ray-samples over the step time.  `--scaling weak` renders N frames per step
(rank r renders tile r of every frame; each GPU traces a whole frame's worth of rays) and rides in `also` at N > 1.  Inputs (pose, tables, weights)
are resident in HBM before the timed region.

Workloads (BASELINE.json configs):  --workload hash    HashNeRF: CuHashEmbedder L16 T2^19 F2 + CuSHEncoder(4) + NeRFSmall
                                     --workload classic PE(10)/PE(4) + NeRF 8x256
Rank 0 prints ONE compact JSON line (< 4 KB: benchlib/report.py); the full record goes to bench_detail.json and to stderr.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.costs import H, W, NS, NI, UNITS_PER_RAY, executed_per_ray, colour_only_per_ray      # noqa: E402
from benchlib import report                                                                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="hash", choices=["hash", "classic"])
    ap.add_argument("--precision", default=None, choices=["f16", "f16x3", "f32"],
                    help="MLP arithmetic: f16 = matrix cores, fp16 operands; f16x3 = matrix cores, hi+lo fp16 operand pairs (fp32-grade); f32 = FMA chains "
                         "(== oracle bitwise).  Default: f16x3 (fp32-grade pixels; the reference computes in fp32)")
    ap.add_argument("--no-settle", action="store_true", help="skip the settling phase before the warmup steps (counter passes: tools/gpu_pmc*.sh count the frames of a run)")
    ap.add_argument("--no-isolated", action="store_true", help="skip the single-lane pass behind roofline.isolated")
    ap.add_argument("--no-also", action="store_true", help="skip the short secondary measurements (other precision, classic workload) at N = 1")
    ap.add_argument("--hash-mode", default="cu", choices=["cu", "ngp"])
    ap.add_argument("--chunk", type=int, default=0, help="rays per RenderRays call (0 = workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the whole-frame render in the NRF_PREC_F32 parity mode after the timed region (profiling runs: keeps the kernel summary to the timed kernels)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all-gather even at world size 1 (self-test of the N > 1 code path)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo lets N ranks SHARE one GPU (a rehearsal of the N > 1 code path on a one-GPU box: RCCL refuses two ranks on a device)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong (default at N > 1, BASELINE config 4): ONE frame per step split over the N ranks; weak: N frames per step, every GPU traces a whole frame's worth of rays")
    ap.add_argument("--collective", default=None, choices=["torch", "cabi"],
                    help="the per-step all-gather: nrf_allgather_tiles (RCCL behind the C ABI, what a C++ host calls; default with --backend nccl, falls back to torch loudly if "
                         "its communicator cannot be created) or torch.distributed (RCCL through PyTorch; the only one with --backend gloo)")
    ap.add_argument("--lanes", default="auto", choices=["auto", "1", "2", "3", "4"],
                    help="streams of the library's Chunk loop (nrf_set_render_lanes); auto: 1 against 2 measured before the warmup steps, the faster one is timed (NRF_RENDER_LANES pins it)")
    ap.add_argument("--overflow-policy", default="deferred", choices=["auto", "deferred", "ignore"],
                    help="nrf_render_params.overflow_policy of the timed frames: deferred (default) keeps the frame calls asynchronous -- the chunks' non-finite words are read by the "
                         "next call and, after the timed region, by nrf_renderer_nonfinite (reported as nonfinite_chunks); auto = one host read-back at the end of every frame call "
                         "(+ an NRF_PREC_F32 re-render of flagged chunks), what a host that renders single frames gets by default")
    ap.add_argument("--dense-mb", type=float, default=-1, help="override the baked dense-level budget of the hash fast path (MB)")
    ap.add_argument("--kernel-stats", default="profiles/round6/r8m_single_lane_kernel_stats.csv",
                    help="named in roofline.kernel_stats: the committed rocprofv3 --kernel-trace --stats summary of the single-lane pass the roofline re-derives from")
    args = ap.parse_args()

    if args.precision is None:
        args.precision = "f16x3"
    if args.backend == "gloo" and args.collective == "cabi":
        sys.exit("--collective cabi is RCCL: one rank per GPU (--backend nccl)")
    if args.collective is None:
        args.collective = "cabi" if args.backend == "nccl" else "torch"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        from benchlib.launch import spawn_ranks
        sys.exit(spawn_ranks(args.gpus, os.path.abspath(__file__)))                 # plain `python bench.py --gpus N`: this process becomes the launcher and never touches the GPU
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if os.environ.get("NRF_BENCH_TEST_FAIL_RANK") == str(rank):
        sys.exit(3)                                      # test hook: a rank that dies at start-up (tests/: the launcher must report it, not hang)
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"   # N = 1: the two coincide (one frame per step), reported as "weak" per the driver's contract

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from benchlib.cpu import cpu_baseline
        cpu = cpu_baseline(args.workload)        # before the GPU is initialised (it may spawn a child process)

    import torch
    import torch.distributed as dist
    from nerfpp_amd import _lib as L, scene
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    torch.cuda.set_device(local if args.backend == "nccl" else local % max(torch.cuda.device_count(), 1))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":           # bind the process group to this rank's device up front (no "guessing device ID" on heterogeneous rank -> GPU maps)
            dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    from nerfpp_amd.dist import TileShard, TileComm
    if use_dist and args.backend != "nccl" and world > 1:
        # a rehearsal: the ranks SHARE one GPU.  Two processes with two lanes each is four streams' worth of kernels time-sliced between two contexts (102 ms per step
        # against 29 with one lane per process): one lane per process there
        args.lanes = "1"

    prec = {"f16": L.NRF_PREC_F16_MFMA, "f16x3": L.NRF_PREC_F16_SPLIT, "f32": L.NRF_PREC_F32}[args.precision]
    if args.workload == "hash":
        sc = scene.make_hash_scene(mode=args.hash_mode)
        if args.dense_mb >= 0 and args.hash_mode == "cu":
            sc["embedder"].set_dense_budget(int(args.dense_mb * (1 << 20)))
        # 65 536 rays per RenderRays call: same-call A/B over 49 152 / 65 536 / 131 072 gives 21.3-21.6 / 21.4-21.7 / 21.8 ms per frame (docs/history/profiles/round4/r4G_*), and every
        # slow-box outlier of the round (frames of 45-130 ms on some boxes, r4d_lane_sweep_first_box.log, r4E_*) was at chunks of 98 304 rays and more
        chunk = args.chunk or 65536
    else:
        sc = scene.make_classic_scene()
        chunk = args.chunk or 8192
    renderer = sc["renderer"]
    rp = scene.lego_render_params(sc["bbox"], NS, NI, chunk, prec)
    rp.OverflowPolicy = {"auto": L.NRF_OVERFLOW_AUTO, "deferred": L.NRF_OVERFLOW_DEFERRED, "ignore": L.NRF_OVERFLOW_IGNORE}[args.overflow_policy]
    K = scene.lego_K(H, W)
    shard = TileShard(H, W, rank, world, force_collective=args.force_dist)
    from benchlib.steps import FrameStepper
    # the per-step all-gather: by default the one a C++ host calls -- nrf_allgather_tiles, RCCL behind the C ABI.  If its communicator cannot be created on ANY rank, all
    # ranks fall back (in this process, loudly) to torch.distributed's RCCL; the line says which one was timed
    comm = None
    collective_note = None
    if use_dist and args.collective == "cabi":
        err = ""
        try:
            comm = TileComm(rank, world, timeout_s=120.0)
        except Exception as e:
            err = str(e)
        ok = torch.tensor([0.0 if comm is None else 1.0], device="cuda")
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() != 1.0:
            print(f"[bench] rank {rank}: nrf_comm_create failed on some rank ({err or 'a peer'}): FALLING BACK to torch.distributed all_gather_into_tensor", file=sys.stderr, flush=True)
            if comm is not None:
                comm.close()
            comm = None
            args.collective = "torch"
            collective_note = "cabi requested, nrf_comm_create failed: torch.distributed timed instead" + (f" ({err[:80]})" if err else "")
    import ctypes as C
    NPROF = len(L.NRF_PROF_NAMES)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    settled = [False]
    from benchlib import timing
    lib = L.lib()
    dist_dev = "cuda" if args.backend == "nccl" else "cpu"

    def agree_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], device=dist_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def set_lanes(n):
        L.check(lib.nrf_set_render_lanes(int(n)))

    def settle(step, drain, budget_s=10.0, floor_s=1.5):
        """Before the first measurement of the process, outside every timed region and before the W warmup steps: whole steps, each synchronised, until three in a
        row are within 10 % of the fastest one seen and at least `floor_s` of them have run (at most `budget_s`).  Some boxes of the pool start a process at a
        fraction of the clock -- the default frame at 97 instead of 22 ms per step for the first seconds of load (docs/history/profiles/round4/r4E_*) -- and W = 1-2 warmup
        steps do not outlast that.  Every rank runs the same number of steps (the decision to stop is taken together)."""
        if settled[0] or args.no_settle:
            return
        settled[0] = True
        t_begin = time.perf_counter(); best = float("inf"); good = 0
        while True:
            sync(); t0 = time.perf_counter()
            step(); drain(); sync()
            dt = time.perf_counter() - t0
            best = min(best, dt)
            good = good + 1 if dt <= 1.10 * best else 0
            el = time.perf_counter() - t_begin
            stop = 1.0 if ((good >= 3 and el >= floor_s) or el >= budget_s) else 0.0
            if use_dist:
                t = torch.tensor([stop], device=dist_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)          # all ranks stop together (each step holds a collective)
                stop = float(t.item())                            # every rank's own clock passes the budget, so the minimum becomes 1 on all of them in the same iteration
            if stop:
                break

    def make_steps(scaling):
        """The step of `scaling` (benchlib/steps.py) -> (step, drain, poses, the list of host seconds this rank spent inside each Render call, render_tiles)."""
        # (overlap with RCCL only: gloo moves the tiles through the host on a helper thread and is slower issued that way -- 38.8 vs 29.4 ms per step in the 2-rank rehearsal)
        fs = FrameStepper(renderer, rp, K, H, W, shard, scaling, world, comm=comm, overlap=args.backend == "nccl")
        return fs.step, fs.drain, fs.poses, fs.host, fs.render_tiles

    lanes_info = dict(requested=args.lanes, chosen=None, ms_per_step_by_lanes=None)

    def timed_run(scaling, steps, warmup, pick_lanes=False):
        """W untimed + K timed steps of `scaling`, the library's per-kernel event bracketing OFF (benchlib/timing.py refuses otherwise); returns (seconds = max over
        ranks, frames of the last step, poses, host seconds this rank spent inside Render calls, render_tiles)."""
        step, drain, poses_, host, render_tiles = make_steps(scaling)
        settle(step, drain)
        if pick_lanes and lanes_info["chosen"] is None:
            if args.lanes == "auto":
                # 1 lane against 2, measured here: on some boxes the second lane costs more than it hides (round 4's driver box: the two-lane default was slower than one lane)
                win, by = timing.choose_lanes(lib, set_lanes, step, drain, sync, agree_max)
                lanes_info.update(chosen=win, ms_per_step_by_lanes={str(k): round(v, 3) for k, v in by.items()})
            else:
                set_lanes(int(args.lanes))
                lanes_info.update(chosen=int(args.lanes))
        for _ in range(warmup):
            step()
        drain()
        del host[:]
        dt, frames_ = timing.timed_region(lib, step, drain, sync, steps)
        return agree_max(dt), frames_, poses_, list(host), render_tiles

    def profiled_pass(scaling, steps, lanes):
        """A short pass with the per-kernel event bracketing ON (never part of `value`): per-kernel HIP-event totals on the launch streams with `lanes` lanes."""
        set_lanes(lanes)
        step, drain, _, _, _ = make_steps(scaling)
        step(); drain(); sync()
        ms = (C.c_double * NPROF)(); cnt = (C.c_int64 * NPROF)()
        lib.nrf_profile_enable(1)
        lib.nrf_profile_read(ms, cnt, 1)
        try:
            sync(); t0 = time.perf_counter()
            for _ in range(steps):
                step()
            drain(); sync()
            dt = time.perf_counter() - t0
            lib.nrf_profile_read(ms, cnt, 1)
        finally:
            lib.nrf_profile_enable(0)
            set_lanes(lanes_info["chosen"] or 2)
        return dict(dt=agree_max(dt), ms=ms, cnt=cnt, steps=steps, lanes=lanes)

    forced_env = os.environ.get("NRF_RENDER_LANES")
    if forced_env and args.lanes == "auto":
        args.lanes = forced_env                  # profiling scripts pin the lane count through the environment
        lanes_info["requested"] = "env:" + forced_env
    elapsed, frames, poses, host_s, render_tiles = timed_run(args.scaling, args.steps, args.warmup, pick_lanes=True)
    lanes_timed = lanes_info["chosen"] or lib.nrf_get_render_lanes()
    nframes = len(poses)
    # at N > 1 the other scaling mode rides along in `also` (every rank takes part; fewer steps)
    other = None
    if world > 1 and not args.no_also:
        o_scaling = "weak" if args.scaling == "strong" else "strong"
        o_steps = max(2, args.steps // 2)
        o_dt, o_frames, o_poses, o_host, _ = timed_run(o_scaling, o_steps, 1)
        other = dict(scaling=o_scaling, frames_per_step=len(o_poses), steps=o_steps, ms_per_step=o_dt / o_steps * 1e3,
                     value=len(o_poses) * H * W * UNITS_PER_RAY * o_steps / o_dt, unit="ray-samples/s",
                     host_ms_per_tile=timing.median(o_host) * 1e3, finite=bool(torch.isfinite(o_frames).all()))
    # untimed cross-check of the other collective implementation on the same tiles (N > 1: the C-ABI all-gather a C++ host calls vs torch.distributed's).
    # It runs on a helper thread with a deadline so that nothing it does can cost the run its result line.
    collective_check = None
    stuck = False
    if use_dist and args.backend == "nccl":
        import threading
        box = {}

        def cross_check():
            try:
                torch.cuda.set_device(local)                       # the current device is per thread
                tiles = render_tiles()
                a = shard.all_gather_frames(tiles)
                c2 = comm if comm is not None else TileComm(rank, world)
                b = c2.all_gather_frames(torch.stack([t.reshape(shard.rows, W, 3) for t in tiles], 0), H)
                torch.cuda.synchronize()
                same = torch.tensor([1.0 if torch.equal(a.reshape(b.shape), b) else 0.0], device="cuda")
                dist.all_reduce(same, op=dist.ReduceOp.MIN)
                box["r"] = ("nrf_allgather_tiles (C ABI, RCCL) == torch.distributed all_gather_into_tensor on every rank" if same.item() == 1.0
                            else "MISMATCH between the two collectives")
            except Exception as e:
                box["r"] = f"nrf_allgather_tiles cross-check failed: {e}"
        th = threading.Thread(target=cross_check, daemon=True)
        th.start()
        th.join(timeout=90.0)
        collective_check = box.get("r", "nrf_allgather_tiles cross-check did not finish within 90 s")
        stuck = th.is_alive()                                        # a stuck collective: report what was timed and leave without tearing the group down

    # data-parallel TRAINING step at N > 1 (every rank takes part; `also`): the gradient all-reduce behind the C ABI.  On a helper thread with a deadline, like the cross-check
    dp_train = None
    if use_dist and world > 1 and not args.no_also and not stuck:
        import threading
        box2 = {}

        def dp_job():
            try:
                torch.cuda.set_device(local if args.backend == "nccl" else local % max(torch.cuda.device_count(), 1))
                from benchlib import extras as _ex
                box2["r"] = _ex.dp_train_step_measurement(L, scene, comm if args.backend == "nccl" else None, rank, world, sync, agree_max)
            except Exception as e:
                box2["r"] = dict(workload="hashnerf_train_step_dp", error=str(e)[:200])
        th2 = threading.Thread(target=dp_job, daemon=True)
        th2.start()
        th2.join(timeout=180.0)
        dp_train = box2.get("r", dict(workload="hashnerf_train_step_dp", error="did not finish within 180 s"))
        stuck = stuck or th2.is_alive()

    # per-kernel times, never from the timed region: a short SINGLE-LANE pass with the event bracketing on (each kernel has the GPU to itself: `roofline`), and, when
    # the timed region ran on more lanes, a short pass on that many (the same kernels sharing the CUs: context)
    isolated = shared = None
    if not args.no_isolated and not stuck:
        i_steps = max(3, min(5, args.steps))
        isolated = profiled_pass(args.scaling, i_steps, 1)
        if lanes_timed > 1:
            shared = profiled_pass(args.scaling, i_steps, lanes_timed)
    units_per_step = nframes * H * W * UNITS_PER_RAY          # over all ranks
    value = units_per_step * args.steps / elapsed


    ranks_seen = None
    if comm is not None:
        ranks_seen = comm.world                       # what RCCL itself reports for the communicator behind the C ABI (nrf_comm_world)
    elif use_dist and args.backend == "nccl":
        ranks_seen = dist.get_world_size()

    if rank == 0:
        from benchlib import extras, roofline as RF
        units_rank = units_per_step / world                                                 # this rank's ray-samples per step
        roof = dict(lanes_timed=lanes_timed, lanes=lanes_info, profile_events_in_timed_region=False)
        if shared is not None:
            # the kernels' launch times with the timed region's lane count (they share the CUs there), from a separate profiled pass
            sprof = RF.prof_table(shared["ms"], shared["cnt"], L.NRF_PROF_NAMES)
            roof["timed"] = RF.kernel_rooflines(sprof, args.workload, args.precision, args.hash_mode, units_rank * shared["steps"])
            roof["kernel_ms_timed"] = sprof
            roof["shared_pass_ms_per_step"] = shared["dt"] / shared["steps"] * 1e3
        if isolated is not None:
            # the same quantities with the Chunk loop on ONE stream (a short pass after the timed region): each kernel has the GPU to itself, so launch time is the
            # kernel's own and the fractions are the kernels'.  THIS is the line's top-level roofline.
            iprof = RF.prof_table(isolated["ms"], isolated["cnt"], L.NRF_PROF_NAMES)
            roof["isolated"] = RF.kernel_rooflines(iprof, args.workload, args.precision, args.hash_mode, units_rank * isolated["steps"])
            roof["isolated_kernel_ms"] = iprof
            roof["isolated_steps"] = isolated["steps"]
            roof["isolated_ms_per_step"] = isolated["dt"] / isolated["steps"] * 1e3
            roof["isolated_kernel_sum_ms_per_step"] = sum(v["ms"] for v in iprof.values()) / isolated["steps"]
        enc = ("CuHashEmbedder" if args.hash_mode == "cu" else "HashEmbedder") + " L16 T2^19 F2 16..512 + " + ("CuSHEncoder" if args.hash_mode == "cu" else "SHEncoder") + " deg4 + NeRFSmall 3x64/4x64"
        detail = {
            "metric": "ray-samples/sec (HIP volume-rendering path, Lego 800x800, N_samples=64+128)",
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            # the arithmetic the path computes in: f16x3 = fp16 matrix cores on (hi, lo) operand pairs, three products, fp32 accumulation (fp32-grade pixels)
            "dtype": {"f16": "f16", "f16x3": "f16x3", "f32": "f32"}[args.precision],
            "data": "synthetic",
            "config": {"workload": ("hashnerf_lego800_64+128" if args.workload == "hash" else "classic_nerf_lego800_64+128"),
                       # 1-based, as BASELINE.md / SURVEY section 8 / DESIGN count them: 2 = classic 800x800 64+128, 3 = HashNeRF on one GPU, 4 = HashNeRF sharded over N GPUs, 5 = LeRF
                       "baseline_config": ((4 if world > 1 else 3) if args.workload == "hash" else 2),
                       "encoder": enc if args.workload == "hash" else "PE(10)/PE(4) + NeRF 8x256 skip4 viewdirs",
                       # CuHashEmbedder / CuSHEncoder are CUDA-only units: their oracle is a line-by-line restatement pinned by known answers, not by a reference run
                       # (DESIGN.md section 2); the reference-pinned LibTorch twin of the same configuration rides in `also`
                       "oracle_pin": "restatement (CUDA-only encoders; reference-pinned twin in also)" if (args.workload == "hash" and args.hash_mode == "cu") else "reference",
                       "frames_per_step": nframes, "rays_per_gpu_per_step": nframes * H * W // world, "ray_samples_per_ray": UNITS_PER_RAY, "chunk": chunk,
                       "parallelism": f"row-tile x{world}" + ((" + " + ("RCCL" if args.backend == "nccl" else "gloo (ranks share one GPU: rehearsal)") +
                                                               " all_gather (" + ("C ABI" if args.collective == "cabi" else "torch.distributed") + ")") if use_dist else "")},
            # value counts the reference's 256 network evaluations per ray; the fine pass's 64 coarse depths reuse the coarse pass's features / outputs
            # (identical results), so the kernels process fewer -- the rooflines price what each kernel really processed
            "executed_evaluations_per_ray": dict(zip(("hash_encode", "fused_mlp", "sigma_only"), executed_per_ray(args.workload, args.precision, args.hash_mode)),
                                                 colour_net_only=colour_only_per_ray(args.workload, args.precision)),
            "rays_per_s": value / UNITS_PER_RAY, "s_per_frame": elapsed / args.steps / nframes * (world if args.scaling == "weak" else 1),
            # wall time rank 0 spent INSIDE Render per tile (one nrf_render_rows call: ~110 asynchronous launches): the unsharded work that bounds strong scaling.
            # MEDIAN over the timed region's calls: the launch work itself (0.4-0.5 ms on an idle queue, tools/scratch/host_time_probe.py).  The mean also holds the calls in
            # which the host, frames ahead of the device, waits for queue space (the device is busy then: not a cost)
            "host_ms_per_tile": timing.median(host_s) * 1e3, "host_ms_per_tile_mean": sum(host_s) / max(len(host_s), 1) * 1e3, "tile_rows": shard.rows,
            "roofline": roof, "ranks_seen_by_rccl": ranks_seen,
            "collective": (("nrf_allgather_tiles (C ABI)" if comm is not None else "torch.distributed") + (": " + collective_note if collective_note else "")) if use_dist else None,
        }
        if other is not None:
            detail["also"] = [other]
        if dp_train is not None:
            detail["also"] = detail.get("also", []) + [dp_train]
        if collective_check is not None:
            detail["collective_check"] = collective_check
        if cpu is not None:
            detail["cpu_baseline"] = cpu
        # parity of what was just timed (cpu_baseline leg, checker use of oracle/): GPU render vs the CPU oracle on identical weights/pose
        try:
            detail["psnr_vs_oracle_db"] = extras.quality_check(sc, renderer, rp, K, poses[0], args)
        except Exception as e:
            detail["psnr_vs_oracle_db"] = f"unavailable: {e}"
        if not args.no_parity:
            try:
                detail["parity_full_frame_vs_f32"] = extras.full_frame_parity(sc, renderer, rp, K, poses[0], args, scene, L)
            except Exception as e:
                detail["parity_full_frame_vs_f32"] = f"unavailable: {e}"
        assert frames.shape[0] == nframes and bool(torch.isfinite(frames).all())
        # the non-finite words of every chunk rendered by this process (matrix-core precisions; include/nerfpp_hip.h, nrf_render_params.overflow_policy)
        nf_flagged, nf_rerendered = renderer.nonfinite()
        detail["overflow_policy"] = args.overflow_policy
        # device memory of the timed configuration on this rank (replicated per rank at N > 1): the hash table, the baked image of the render fast path, the Chunk loop's
        # workspace, and the device's total in use (torch's pool, the library's own allocations, the runtime)
        mem = {}
        if args.workload == "hash":
            tb, bb = C.c_int64(0), C.c_int64(0)
            L.check(lib.nrf_hash_memory_bytes(sc["embedder"]._h, C.byref(tb), C.byref(bb)))
            mem.update(hash_table=int(tb.value), baked_fast_path_image=int(bb.value))
        mem["render_workspace"] = int(renderer._ws.numel()) if getattr(renderer, "_ws", None) is not None else 0
        free_b, total_b = torch.cuda.mem_get_info()
        mem["device_in_use"] = int(total_b - free_b)
        detail["memory_bytes"] = mem
        detail["nonfinite_chunks"] = dict(flagged=nf_flagged, rerendered_in_f32=nf_rerendered)
        assert nf_flagged == 0, "a timed frame produced non-finite network outputs"
        # the gathered frame of the first pose, hashed: equal strings at different N (or launchers) = the sharded render is the single-GPU render bit for bit
        import hashlib
        detail["frame_sha256"] = hashlib.sha256(frames[0].reshape(H, W, 3).contiguous().cpu().numpy().tobytes()).hexdigest()
        if world == 1 and not use_dist and not args.no_also:
            detail["also"] = extras.secondary_measurements(args, scene, L, K, poses[0], sc)
        # The contract is ONE JSON line.  RCCL prints a version banner through C stdio when the process exits, and tearing the process group down is a collective
        # that a slow or already-gone peer can stall: drain what is buffered, print the line, flush -- and in the distributed case every rank then leaves without
        # running tear-down or exit-time printers (everything that was to be measured and checked is in the line / the side file).
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        report.emit(detail, stats_csv=args.kernel_stats)
    if use_dist:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
