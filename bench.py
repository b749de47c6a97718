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
One JSON line is printed by rank 0 (see the keys at the bottom).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
NS, NI = 64, 128
UNITS_PER_RAY = NS + NS + NI          # one shared network, fine pass re-evaluates all depths (NeRFRenderer.h:422,447): the metric's unit count per ray
# What the kernels actually execute per ray on the matrix-core paths: the S coarse depths of the fine set are the coarse pass's own sample points, so their hash
# features (default split mode) or their whole network outputs (coarse pass = whole network in the same arithmetic) are reused -- results unchanged.  `value` keeps
# counting the reference's 256 evaluations per ray (the work the frame stands for); the rooflines below price what each kernel really processed.


def executed_per_ray(workload, precision, hash_mode, coarse_full=False):
    """(hash-encode points, fused-MLP points, sigma-only points) per ray."""
    if precision == "f32":
        return UNITS_PER_RAY, UNITS_PER_RAY, 0
    if workload == "classic":
        if precision == "f16x3" and coarse_full is False:
            return 0, NI, NS                                      # coarse pass: density branch in exact fp32 + colour branch on the exact h8 (sigma_nerf_f32.hip); the fine pass evaluates the 128 new depths
        return 0, NS + NI, 0                                      # whole network on the coarse pass, its outputs reused by the fine pass: 64 + 128 evaluations
    if precision == "f16x3":                                  # coarse pass: sigma net alone (exact fp32), which hands (sigma, geo_feat) to the fine pass
        return NS + NI, NI, NS                                # both encoders: the fine pass keeps the coarse pass's feature columns; whole network on the new samples only
    return NS + NI, NS + NI, 0                                # plain fp16: coarse outputs reused by the fine pass


def colour_only_per_ray(workload, precision):
    """Points per ray at which the fused-MLP kernel runs the colour net alone (HashNeRF default mode: the fine pass's S coarse depths, whose sigma-net output comes
    from the exact coarse kernel)."""
    return NS if (workload == "hash" and precision == "f16x3") else 0

# algorithmic cost per ray-sample (SURVEY.md section 8d / BASELINE.md section 2)
HASH_BYTES_PER_UNIT = 16 * 8 * 2 * 2 + 12 + 64      # table gathers + point in + fp16 features out (standalone encode kernel)
SMALL_FLOP_PER_UNIT = 35072
SMALL_COLOUR_FLOP_PER_UNIT = 2 * ((16 + 15) * 64 + 64 * 64 + 64 * 64 + 64 * 3)      # the colour net alone (NeRF.cpp:383-406): 20 736
SMALL_COLOUR_MFMA_FLOP_PER_UNIT = 72 * 32768 // 32                                   # its 72 of the split kernel's 116 matrix instructions per 32 points
# matrix-core work the NeRFSmall kernel actually issues per point (32-row / 16-k padded tiles; x3 products in split mode, x2 on layer 0)
SMALL_MFMA_FLOP_PER_UNIT = {"f16": 40 * 32768 // 32, "f16x3": 116 * 32768 // 32}
NERF_FLOP_PER_UNIT = 1186816
# the density branch alone (NeRF.cpp:92-108: pts_linears 0..7 with the skip-concat, alpha_linear): 491 264 MAC
NERF_SIGMA_FLOP_PER_UNIT = 2 * (63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256 + 256)
# the coarse pass of the default mode needs sigma only (NeRFRenderer.h:422-428): in 32 -> 64 -> 64 -> 1
SIGMA_FLOP_PER_UNIT = 2 * (32 * 64 + 64 * 64 + 64)
# LeRF head at main.cpp:203-213 sizes: 128 -> 256 -> 33 ; cat[geo32, in128] -> 256 -> 768, bias-free
LERF_FLOP_PER_UNIT = 557568
LERF_SIGMA_FLOP_PER_UNIT = 2 * (128 * 256 + 256 * 33)                        # the density net alone (the coarse pass's exact-fp32 kernel, geo rows included)
# matrix instructions (32x32x16 fp16 = 32 768 flop) the split-precision LeRF passes issue per 32 points: the sigma pass on the new samples (layer 0 on exact-fp16
# features: 2 products, layer 1: 3) and the embedding pass from LE0 on (LE0: 8 tiles x (8 x 2 + 4 x 3), Gram: 8 x 16 x 1); the 256 -> 768 layer runs once per RAY
LERF_SPLIT_MFMA_SIGMA = 8 * 8 * 2 + 2 * 16 * 3
LERF_SPLIT_MFMA_EMBED = 8 * (8 * 2 + 4 * 3) + 8 * 16 * 1                       # LE0 in split precision; the Gram product (a scalar norm per sample) on the hi parts only
LERF_HASH_BYTES_PER_UNIT = 16 * 8 * 8 * 2 + 12 + 16 * 8 * 2                    # CuHash F = 8: 2 048 B of table reads + the point + 256 B of level-major fp16 features
HBM_PEAK = 8.0e12
MFMA_F16_PEAK = 2.5e15
# MI355X_MICROARCH.md, "DVFS give-back" item 1: the chip lowers its clock under matrix load; a tuned bf16 GEMM on random data holds 1.90-1.95 GHz and
# delivers 1 247 TFLOP/s (1 483 on all-zero operands at 2.30 GHz).  The rate the matrix pipes SUSTAIN on real data, as measured by the guide.
MFMA_F16_SUSTAINED_GEMM = 1.247e15
F32_PEAK = 157.3e12
# MI355X_MICROARCH.md, "Indexed rows: gather": uniformly random rows of a table served from the Infinity Cache read at 8.6 TB/s chip-wide
# (16.8-18.8 TB/s when every row is L2-resident, 6.0 TB/s swept from HBM) -- the ceiling of the vector-memory gather path the hash encode runs on
GATHER_PEAK = 8.6e12
GATHER_PEAK_L2 = 16.8e12


def cpu_baseline(workload, seconds_target=12.0):
    """The reference's own CPU renderer (oracle/_ref/ref_driver, kind 'reference') when that binary travelled with the
    repo, else the C oracle ('port'), on a bounded sample of the same workload: image rows of the same camera, >= 4 800 rays.
    LibTorch's intra-op pool is pinned per run (OMP_NUM_THREADS): 8 / 16 / 32 / 64 threads are swept on a 6-row sample and the best
    count then renders the timed sample -- 128 threads on a few thousand rays is an oversubscription artefact, not a baseline."""
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    fam = "hash" if workload == "hash" else "classic"
    if os.path.exists(drv):
        try:
            ncpu = os.cpu_count() or 8

            def run(rows, threads):
                env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
                out = subprocess.run([drv, "bench", fam, str(H), str(rows), str(NS), str(NI), "4096", "1"], capture_output=True, text=True, timeout=600, env=env)
                return json.loads(out.stdout.strip().splitlines()[-1])
            sweep = {}
            for t in sorted({min(t, ncpu) for t in (8, 16, 32, 64)}):
                sweep[t] = run(6, t)
            best_t = max(sweep, key=lambda t: sweep[t]["units_per_s"])
            probe = sweep[best_t]
            rows = int(max(6, min(96, 6 * seconds_target / max(probe["seconds"], 1e-3))))
            r = run(rows, best_t) if rows > 6 else probe
            return dict(value=r["units_per_s"], unit="ray-samples/s", cores=r["threads"], kind="reference",
                        thread_sweep={str(t): round(v["units_per_s"]) for t, v in sweep.items()}, host_cpus=ncpu,
                        sample=f"{r['rays']} rays ({rows} rows of the {H}x{W} frame), {NS}+{NI} samples, Chunk 4096, LibTorch CPU "
                               f"{'HashEmbedder+SHEncoder+NeRFSmall' if fam == 'hash' else 'PE+NeRF 8x256'}, {r['seconds']:.1f} s, "
                               f"{r['threads']} threads (best of the 8/16/32/64 sweep)")
        except Exception as e:  # fall through to the port
            print(f"[bench] reference driver failed ({e}); timing the oracle port instead", file=sys.stderr)
    from oracle import capi as O
    from nerfpp_amd import scene, synth
    bbox = scene.LEGO_BBOX
    if workload == "hash":
        table = scene.synth_hash_table(16, 19, 2, 5000, 0.5)
        blob = np.concatenate([a.reshape(-1) for _, a in scene.synth_linear_stack(scene.small_shapes(), 6000, 1.6, 0.0, {"sigma_net_2": 30.0})])
        model = O.Model(0, blob, bbox=bbox, table_f32=table)
    else:
        blob = np.concatenate([a.reshape(-1) for _, a in scene.synth_linear_stack(scene.nerf_shapes(), 7000, 1.4, 0.1, {"alpha_linear.weight": 40.0})])
        model = O.Model(1, blob, bbox=bbox)
    K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)

    def run(rows):
        o, d, _ = O.get_rays(H, W, K, c2w, row0=H // 2 - rows // 2, rows=rows)
        rays = O.pack_rays(o, d, bbox)
        t0 = time.time()
        O.render_rays(model, rays, NS, NI, O.linspace(0, 1, NS), O.linspace(0, 1, NI), white_bkgr=True)
        return rays.shape[0], time.time() - t0
    n, t = run(6)
    rows = int(max(6, min(96, 6 * seconds_target / max(t, 1e-3))))
    if rows > 6:
        n, t = run(rows)
    return dict(value=n * UNITS_PER_RAY / t, unit="ray-samples/s", cores=O.num_threads(), kind="port",
                sample=f"{n} rays ({rows} rows of the {H}x{W} frame), {NS}+{NI} samples, C oracle with OpenMP, {t:.1f} s")


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, exactly what torch.distributed.run would set), relay rank 0's JSON line, return non-zero if any rank failed.  The parent never
    initialises the GPU and never replaces itself (no exec): it waits for its children."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))      # as torch.distributed.run does: N ranks x all host cores of intra-op threads is oversubscription
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = ""
    deadline = time.time() + float(os.environ.get("NRF_BENCH_TIMEOUT", "1500"))
    try:
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.setdefault("out", procs[0].stdout.read()), daemon=True)
        th.start()
        rc = [None] * n
        while any(c is None for c in rc) and time.time() < deadline:
            for i, pr in enumerate(procs):
                if rc[i] is None:
                    rc[i] = pr.poll()
            if any(c not in (None, 0) for c in rc):
                break                                    # a rank died: its peers would wait in a collective for ever
            time.sleep(0.05)
        th.join(timeout=5.0)
        out0 = box.get("out", "") or ""
    finally:
        for pr in procs:                                 # exact PIDs of the children this process started
            if pr.poll() is None:
                pr.kill()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except Exception:
                pass
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    codes = [pr.returncode for pr in procs]
    if any(c != 0 for c in codes) or not lines:
        print(f"[bench] rank exit codes {codes}" + ("" if lines else "; rank 0 printed no result line"), file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="hash", choices=["hash", "classic"])
    ap.add_argument("--precision", default=None, choices=["f16", "f16x3", "f32"],
                    help="MLP arithmetic: f16 = matrix cores, fp16 operands; f16x3 = matrix cores, hi+lo fp16 operand pairs (fp32-grade); f32 = FMA chains "
                         "(== oracle bitwise).  Default: f16x3 (fp32-grade pixels; the reference computes in fp32)")
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
    ap.add_argument("--collective", default="torch", choices=["torch", "cabi"],
                    help="the per-step all-gather: torch.distributed (RCCL through PyTorch) or nrf_allgather_tiles (RCCL behind the C ABI, what a C++ host calls)")
    ap.add_argument("--dense-mb", type=float, default=-1, help="override the baked dense-level budget of the hash fast path (MB)")
    args = ap.parse_args()

    if args.precision is None:
        args.precision = "f16x3"
    if args.backend == "gloo" and args.collective == "cabi":
        sys.exit("--collective cabi is RCCL: one rank per GPU (--backend nccl)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))                 # plain `python bench.py --gpus N`: this process becomes the launcher and never touches the GPU
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if os.environ.get("NRF_BENCH_TEST_FAIL_RANK") == str(rank):
        sys.exit(3)                                      # test hook: a rank that dies at start-up (tests/: the launcher must report it, not hang)
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"   # N = 1: the two coincide (one frame per step), reported as "weak" per the driver's contract

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    from nerfpp_amd.dist import TileShard, TileComm
    if use_dist and args.backend != "nccl" and world > 1:
        # a rehearsal: the ranks SHARE one GPU.  Two processes with two lanes each is four streams' worth of kernels time-sliced between two contexts (102 ms per step
        # against 29 with one lane per process): one lane per process there
        L.check(L.lib().nrf_set_render_lanes(1))
        os.environ["NRF_RENDER_LANES"] = "1"

    prec = {"f16": L.NRF_PREC_F16_MFMA, "f16x3": L.NRF_PREC_F16_SPLIT, "f32": L.NRF_PREC_F32}[args.precision]
    if args.workload == "hash":
        sc = scene.make_hash_scene(mode=args.hash_mode)
        if args.dense_mb >= 0 and args.hash_mode == "cu":
            sc["embedder"].set_dense_budget(int(args.dense_mb * (1 << 20)))
        chunk = args.chunk or 131072
    else:
        sc = scene.make_classic_scene()
        chunk = args.chunk or 8192
    renderer = sc["renderer"]
    rp = scene.lego_render_params(sc["bbox"], NS, NI, chunk, prec)
    K = scene.lego_K(H, W)
    shard = TileShard(H, W, rank, world, force_collective=args.force_dist)
    comm = TileComm(rank, world) if (use_dist and args.collective == "cabi") else None
    import ctypes as C
    NPROF = len(L.NRF_PROF_NAMES)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(scaling, steps, warmup, profile=False):
        """W untimed + K timed steps of `scaling`; returns (seconds = max over ranks, frames of the last step, poses, host seconds this rank spent inside Render
        calls, per-kernel HIP-event totals)."""
        # frames of one step: poses on the reference's test orbit (pose_spherical(theta, -30, 4), theta step 9 degrees); strong scaling: one frame
        nfr = 1 if scaling == "strong" else world
        poses_ = [scene.pose_spherical(-180.0 + 9.0 * k, -30.0, 4.0) for k in range(nfr)]
        host = [0.0]

        def render_tiles():
            t_h = time.perf_counter()
            tiles = [renderer.Render(H, W, K, rp, c2w=c2w, row0=shard.row0, rows=shard.rows).Outputs.RGBMap for c2w in poses_]   # one nrf_render_rows call each
            host[0] += time.perf_counter() - t_h
            return tiles

        pending = [None]

        def step():
            # the all-gather of step k is issued asynchronously and completed (stream order, no host wait) at step k + 1: the next frame's kernels are not held behind
            # a latency-bound collective -- what a renderer of consecutive frames does
            # (RCCL only: gloo moves the tiles through the host on a helper thread and is slower issued that way -- 38.8 vs 29.4 ms per step in the 2-rank rehearsal)
            tiles = render_tiles()
            ov = args.backend == "nccl"
            if comm is not None:
                out, work = comm.all_gather_frames(torch.stack([t.reshape(shard.rows, W, 3) for t in tiles], 0), H, overlap=True)
            elif ov:
                out, work = shard.all_gather_frames(tiles, overlap=True)     # [frames, H, W, 3] on every rank; identity at N = 1
            else:
                out, work = shard.all_gather_frames(tiles), None
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = work
            return out

        def drain():
            if pending[0] is not None:
                pending[0].wait()
                pending[0] = None

        for _ in range(warmup):
            step()
        drain()
        ms = (C.c_double * NPROF)(); cnt = (C.c_int64 * NPROF)()
        if profile:
            L.lib().nrf_profile_enable(1)
            L.lib().nrf_profile_read(ms, cnt, 1)
        host[0] = 0.0
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            frames_ = step()
        drain()                                   # the last frame's collective, inside the timed region
        sync()
        dt = time.perf_counter() - t0
        if profile:
            L.lib().nrf_profile_read(ms, cnt, 1)
            L.lib().nrf_profile_enable(0)
        if use_dist:
            t = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, frames_, poses_, host[0], (ms, cnt), render_tiles

    elapsed, frames, poses, host_s, (ms, cnt), render_tiles = timed_run(args.scaling, args.steps, args.warmup, profile=True)
    nframes = len(poses)
    # at N > 1 the other scaling mode rides along in `also` (every rank takes part; fewer steps)
    other = None
    if world > 1 and not args.no_also:
        o_scaling = "weak" if args.scaling == "strong" else "strong"
        o_steps = max(2, args.steps // 2)
        o_dt, o_frames, o_poses, o_host, _, _ = timed_run(o_scaling, o_steps, 1)
        other = dict(scaling=o_scaling, frames_per_step=len(o_poses), steps=o_steps, ms_per_step=o_dt / o_steps * 1e3,
                     value=len(o_poses) * H * W * UNITS_PER_RAY * o_steps / o_dt, unit="ray-samples/s",
                     host_ms_per_tile=o_host / o_steps / len(o_poses) * 1e3, finite=bool(torch.isfinite(o_frames).all()))
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

    # rooflines of the kernels one at a time: a short single-lane pass (every rank takes part), see `roofline.isolated`
    isolated = None
    if not args.no_isolated and not stuck and os.environ.get("NRF_RENDER_LANES", "2") != "1":
        L.check(L.lib().nrf_set_render_lanes(1))
        i_steps = max(3, min(5, args.steps))
        i_dt, _, _, _, (i_ms, i_cnt), _ = timed_run(args.scaling, i_steps, 1, profile=True)
        L.check(L.lib().nrf_set_render_lanes(2))
        isolated = dict(dt=i_dt, ms=i_ms, cnt=i_cnt, steps=i_steps)
    units_per_step = nframes * H * W * UNITS_PER_RAY          # over all ranks
    value = units_per_step * args.steps / elapsed

    if rank == 0:
        def rooflines(ms, cnt, steps_):
            """(kernel_ms, roofline) from the HIP-event totals of `steps_` timed steps."""
            prof = {n: dict(ms=ms[i], launches=int(cnt[i])) for i, n in enumerate(L.NRF_PROF_NAMES)}
            # per-launch figures from HIP events on the launch stream; hash workload: the two candidates for `dominant` are the hash encode (HBM) and the fused MLP (MFMA)
            if args.workload == "hash":
                k = prof["hash"]
                units_total = units_per_step * steps_ / world                     # this rank's units over the timed region (256 per ray)
                ex_hash, ex_mlp, ex_sigma = executed_per_ray(args.workload, args.precision, args.hash_mode)
                units_per_launch = units_total * ex_hash / UNITS_PER_RAY / max(k["launches"], 1)          # points the hash kernel really encoded
                dur = k["ms"] * 1e-3 / max(k["launches"], 1)
                achieved = units_per_launch * HASH_BYTES_PER_UNIT / max(dur, 1e-12)
                traffic, traffic_src = pmc_traffic("hash_encode (k_hash_cu_lm)", units_per_launch)
                # The baked pyramid is read through L2 / Infinity Cache (PMC: a third of the requested bytes reach HBM), so the bound is the vector-memory GATHER
                # path, not HBM: `achieved` prices the algorithmic bytes (588 B per unit) against the guide's measured ceiling for cache-resident
                # gathers; hbm_frac is what the HBM counters saw, against the 8 TB/s SURVEY 8d names.
                # achieved = algorithmic bytes (588 B per point the kernel encoded) / kernel time, against the 8 TB/s HBM peak.  The baked pyramid is read
                # through L2 / Infinity Cache (PMC: a third of the requested bytes reach HBM), which is how `frac` can exceed 1; the ceilings of the vector-memory
                # gather path the kernel really runs on ride along (frac_of_l2_gather_ceiling / over_infinity_cache_gather_rate), and hbm_frac is what the HBM counters saw.
                roof = dict(bound="hbm", kernel="hash_encode", achieved=achieved / 1e9, peak=HBM_PEAK / 1e9,
                            unit="GB/s", frac=achieved / HBM_PEAK, frac_of_l2_gather_ceiling=achieved / GATHER_PEAK_L2, over_infinity_cache_gather_rate=achieved / GATHER_PEAK,
                            hbm_frac=(traffic / max(dur, 1e-12) / HBM_PEAK) if traffic else None, algorithmic_over_hbm_peak=achieved / HBM_PEAK,
                            traffic=traffic, traffic_source=traffic_src, launches=k["launches"], avg_launch_ms=dur * 1e3,
                            units_per_launch=units_per_launch, bytes_per_unit=HASH_BYTES_PER_UNIT,
                            peak_source="peak = HBM3E 8 TB/s (MI355X_MICROARCH.md; SURVEY 8d).  Gather path, same guide, 'Indexed rows: gather': 16.8-18.8 TB/s with the rows resident in the XCD's L2 (frac_of_l2_gather_ceiling); "
                                        "8.6 TB/s when the rows come from the Infinity Cache (38 MB table), 6.0 TB/s swept from HBM.  The 1.1 GB pyramid's coarse levels are L2-resident, "
                                        "its fine levels are not: the kernel runs between the two rates (over_infinity_cache_gather_rate)")
                # a model that fits the ablations, not a documented figure: a gather whose 64 lanes fall into 64 different lines holds the CU's vector L1 for ~64 clocks; a hash lookup is
                # two 16-byte gathers per point and level (8 corners = two quads), so a launch cannot finish before units x 32 lookups / (256 CUs x clock)
                roof["l1_tag_lookup_model"] = dict(line_lookups_per_unit=32, floor_ms_per_launch=[units_per_launch * 32 / (256 * 2.4e9) * 1e3, units_per_launch * 32 / (256 * 2.1e9) * 1e3],
                                                    floor_clock_ghz=[2.4, 2.1], frac_of_floor_at_2p1_ghz=units_per_launch * 32 / (256 * 2.1e9) / max(dur, 1e-12),
                                                    note="ablation builds (DESIGN section 9): gathers alone 7.8 ms, vector work alone 5.6 ms, whole kernel 8.2-8.3 ms per frame")
                mk = prof["mlp"]
                mdur = mk["ms"] * 1e-3
                mlp_peak = MFMA_F16_PEAK if args.precision != "f32" else F32_PEAK
                sk = prof["sigma"]
                # points the fused MLP kernel really processed (see executed_per_ray)
                mlp_units = units_total * ex_mlp / UNITS_PER_RAY                       # whole-network launches (the kernel instance rocprofv3 lists as k_mlp_small_mfma<..., GEOIN = false>)
                mupl = mlp_units / max(mk["launches"], 1)
                mtraffic, mtraffic_src = pmc_traffic("mlp_small (k_mlp_small_mfma)", mupl)
                mflop = mlp_units * SMALL_FLOP_PER_UNIT
                mroof = dict(bound="mfma", kernel="mlp_small", achieved=mflop / max(mdur, 1e-12) / 1e12, peak=mlp_peak / 1e12,
                             unit="TFLOP/s", frac=mflop / max(mdur, 1e-12) / mlp_peak, traffic=mtraffic, traffic_source=mtraffic_src,
                             launches=mk["launches"], avg_launch_ms=mdur * 1e3 / max(mk["launches"], 1), units_per_launch=mupl, flop_per_unit=SMALL_FLOP_PER_UNIT)
                ck = prof["mlp_colour"]
                if ck["launches"]:
                    # the colour-net-only launches of the fine pass (its S coarse depths; sigma and geo_feat come from the exact coarse kernel): own slot, own kernel instance (GEOIN = true)
                    cdur = ck["ms"] * 1e-3
                    col_units = units_total * colour_only_per_ray(args.workload, args.precision) / UNITS_PER_RAY
                    mroof["colour_only"] = dict(bound="mfma", kernel="mlp_small, colour net alone", achieved=col_units * SMALL_COLOUR_FLOP_PER_UNIT / max(cdur, 1e-12) / 1e12, peak=mlp_peak / 1e12,
                                                unit="TFLOP/s", frac=col_units * SMALL_COLOUR_FLOP_PER_UNIT / max(cdur, 1e-12) / mlp_peak, launches=ck["launches"],
                                                avg_launch_ms=cdur * 1e3 / ck["launches"], units_per_launch=col_units / ck["launches"], flop_per_unit=SMALL_COLOUR_FLOP_PER_UNIT,
                                                mfma_issued_frac=col_units * SMALL_COLOUR_MFMA_FLOP_PER_UNIT / max(cdur, 1e-12) / mlp_peak)
                if args.precision in SMALL_MFMA_FLOP_PER_UNIT:     # issued matrix-core flops (padding + the 3 products of the split mode) / peak
                    missued = mlp_units * SMALL_MFMA_FLOP_PER_UNIT[args.precision]
                    mroof["mfma_issued_frac"] = missued / max(mdur, 1e-12) / mlp_peak
                    mroof["mfma_issued_vs_sustained_gemm"] = missued / max(mdur, 1e-12) / MFMA_F16_SUSTAINED_GEMM
                    mroof["sustained_note"] = ("mfma_issued_vs_sustained_gemm = issued matrix-core flop/s over the 1 247 TFLOP/s a tuned bf16 GEMM holds on random data (guide, DVFS give-back: "
                                               "the chip lowers its clock under matrix load); measured on the classic split kernel: its cycle count does not change on all-zero weights while its clock does (DESIGN section 9)")
                    mroof["note"] = ("achieved / frac price the ALGORITHMIC 35 072 flop per unit; the split-precision mode issues three fp16 products per algorithmic one "
                                     "to deliver fp32-grade pixels (north_star: within 1e-4), mfma_issued_frac is the matrix pipe's own utilisation") if args.precision == "f16x3" else \
                                    "achieved / frac price the algorithmic 35 072 flop per unit; mfma_issued_frac includes the zero padding of the 16- and 3-wide layers"
                    busy = pmc_mfma_busy("mlp_small (k_mlp_small_mfma)", args.precision)
                    if busy:
                        mroof["mfma_busy_frac_of_active_cycles"] = busy
                    try:
                        mroof["clock_ghz_measured"] = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))["mlp_small (k_mlp_small_mfma)"]["clock_ghz_measured"][args.precision]
                    except Exception:
                        pass
                sroof = None
                if sk["launches"]:
                    sdur = sk["ms"] * 1e-3
                    s_units = units_total * ex_sigma / UNITS_PER_RAY
                    straffic, straffic_src = pmc_traffic("sigma_small_f32 (k_sigma_small_f32)", s_units / sk["launches"])
                    sroof = dict(traffic=straffic, traffic_source=straffic_src, mfma_busy_frac_of_active_cycles=pmc_mfma_busy("sigma_small_f32 (k_sigma_small_f32)", "f16x3"),
                                 bound="mfma (fp32, v_mfma_f32_32x32x2_f32)", kernel="sigma_small_f32 (coarse pass: sigma net alone, exact fp32)", achieved=s_units * SIGMA_FLOP_PER_UNIT / max(sdur, 1e-12) / 1e12,
                                 peak=F32_PEAK / 1e12, unit="TFLOP/s", frac=s_units * SIGMA_FLOP_PER_UNIT / max(sdur, 1e-12) / F32_PEAK, launches=sk["launches"],
                                 avg_launch_ms=sdur * 1e3 / sk["launches"], units_per_launch=s_units / sk["launches"], flop_per_unit=SIGMA_FLOP_PER_UNIT,
                                 note="flop_per_unit prices the exact-fp32 sigma chain (32 -> 64 -> 64 -> 1); the kernel also forms the 15 geo_feat rows of the last layer for the fine pass "
                                      "in split fp16 (12 of its 204 matrix instructions per 32 points) and stores them as the colour net's operand fragment (64 B per point)")
                # the roofline object describes the kernel that took the most time in THIS run; the others ride along under their own keys
                # (the NeRFSmall kernel's two instances -- whole network / colour net alone -- count as one kernel here; each keeps its own flops and launch time in the object)
                cands = [(k["ms"], "hash", roof), (mk["ms"] + prof["mlp_colour"]["ms"], "mlp", mroof)] + ([(sk["ms"], "sigma", sroof)] if sroof else [])
                cands.sort(key=lambda c: -c[0])
                roof = dict(cands[0][2])
                for _, name, r_ in cands[1:]:
                    roof[name] = r_
            else:
                k = prof["mlp"]
                dur_total = k["ms"] * 1e-3
                ex_mlp = executed_per_ray(args.workload, args.precision, args.hash_mode, coarse_full=False)[1]
                exec_units = units_per_step * steps_ / world * ex_mlp / UNITS_PER_RAY          # network evaluations this rank's kernel really ran
                flops = exec_units * NERF_FLOP_PER_UNIT
                peak = MFMA_F16_PEAK if args.precision != "f32" else F32_PEAK
                upl = exec_units / max(k["launches"], 1)
                traffic, traffic_src = pmc_traffic("mlp_nerf_split (k_mlp_nerf_split)" if args.precision == "f16x3" else "mlp_nerf (k_mlp_nerf_mfma)", upl)
                roof = dict(bound="mfma", kernel="mlp_nerf" + ("_split" if args.precision == "f16x3" else ""), achieved=flops / max(dur_total, 1e-12) / 1e12, peak=peak / 1e12, unit="TFLOP/s",
                            frac=flops / max(dur_total, 1e-12) / peak, traffic=traffic, traffic_source=traffic_src, launches=k["launches"], units_per_launch=upl,
                            avg_launch_ms=dur_total * 1e3 / max(k["launches"], 1), flop_per_unit=NERF_FLOP_PER_UNIT,
                            note="achieved / frac price the ALGORITHMIC 1 186 816 flop per unit of NeRFImpl::forward as written (11 linear layers); the kernel runs "
                                 "feature_linear and views_linears_0 (no activation in between) as one pre-multiplied affine layer, 10.6 % fewer matrix instructions")
                if args.precision == "f16x3":      # three fp16 products per algorithmic one (hi + lo operand pairs)
                    roof["mfma_issued_frac"] = 3.0 * (1058 * 32768 / 32) * exec_units / max(dur_total, 1e-12) / peak
                    roof["mfma_issued_vs_sustained_gemm"] = 3.0 * (1058 * 32768 / 32) * exec_units / max(dur_total, 1e-12) / MFMA_F16_SUSTAINED_GEMM
                    roof["note"] += "; split precision issues 3 x 1 058 matrix instructions per 32 points (mfma_issued_frac) to deliver fp32-grade pixels"
                busy = pmc_mfma_busy("mlp_nerf_split (k_mlp_nerf_split)" if args.precision == "f16x3" else "mlp_nerf (k_mlp_nerf_mfma)", args.precision)
                if busy:
                    roof["mfma_busy_frac_of_active_cycles"] = busy
            return prof, roof

        prof, roof = rooflines(ms, cnt, args.steps)
        if isolated is not None:
            # the same quantities with the Chunk loop on ONE stream (a short pass after the timed region): there each kernel has the GPU to itself, so launch time is the
            # kernel's own and the fractions are the kernels' -- in the timed region two chunks' kernels share the CUs and a launch's duration includes the sharing
            iprof, iroof = rooflines(isolated["ms"], isolated["cnt"], isolated["steps"])
            iroof["kernel_ms"] = iprof
            iroof["ms_per_step"] = isolated["dt"] / isolated["steps"] * 1e3
            iroof["note_isolated"] = ("nrf_set_render_lanes(1): chunks one after another on the caller's stream, %d steps right after the timed region; the headline's timed region runs "
                                      "consecutive chunks on two streams (one chunk's gather-bound encode beside another's matrix-bound network)" % isolated["steps"])
            roof["isolated"] = iroof
            roof["note_overlap"] = ("timed region: two lanes -- kernels of two chunks share the CUs, so avg_launch_ms (and every fraction derived from it) includes the sharing; "
                                    "`isolated` holds the same figures measured one kernel at a time")
        line = {
            "metric": "ray-samples/sec (HIP volume-rendering path, Lego 800x800, N_samples=64+128)",
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": {"f16": "f16 MFMA (fp32 accumulate) MLP; ", "f16x3": "split-f16 MFMA (hi+lo operand pairs, 3 products, fp32 accumulate: fp32-grade) MLP; ",
                      "f32": "f32 MLP; "}[args.precision] +
                     ("fp16 hash table, fp32 blend" if (args.workload == "hash" and args.hash_mode == "cu") else "f32 encoders") + "; f32/f64 compositing",
            "data": "synthetic",
            "config": {"workload": ("hashnerf_lego800_64+128" if args.workload == "hash" else "classic_nerf_lego800_64+128"),
                       "baseline_config": (2 if args.workload == "hash" else 1),
                       "encoder": (("CuHashEmbedder" if args.hash_mode == "cu" else "HashEmbedder") + " L16 T2^19 F2 16..512 + " +
                                   ("CuSHEncoder" if args.hash_mode == "cu" else "SHEncoder") + " deg4 + NeRFSmall 3x64/4x64") if args.workload == "hash"
                       else "PE(10)/PE(4) + NeRF 8x256 skip4 viewdirs",
                       "oracle_pin": ("CuHashEmbedder / CuSHEncoder are CUDA-only units: the oracle for them is a line-by-line restatement pinned by hand-computed known answers, not by a "
                                      "reference run.  What a real CUDA build could change was measured by modelling it in the oracle (tests/test_oracle_golden.py, sensitivity study): nvcc's FMA "
                                      "contraction moves no pixel by more than 2e-5; one ulp of the level scales, which the reference computes on the device with CUDA's exp2f / log2f, moves the "
                                      "median pixel by 3-4e-4 -- nrf_hash_set_level_scales takes a CUDA build's values.  The reference-pinned encoders are the LibTorch twin in `also`")
                       if (args.workload == "hash" and args.hash_mode == "cu") else "reference-pinned (goldens from the compiled reference)",
                       "frames_per_step": nframes, "rays_per_gpu_per_step": nframes * H * W // world, "ray_samples_per_ray": UNITS_PER_RAY, "chunk": chunk,
                       "parallelism": f"row-tile x{world}" + ((" + " + ("RCCL" if args.backend == "nccl" else "gloo (ranks SHARING one GPU: a rehearsal of the N > 1 code path)") +
                                                               " all_gather (" + ("nrf_allgather_tiles, C ABI" if args.collective == "cabi" else "torch.distributed") + ")") if use_dist else "")},
            "executed_evaluations_per_ray": dict(zip(("hash_encode", "fused_mlp", "sigma_only"), executed_per_ray(args.workload, args.precision, args.hash_mode)),
                                                 colour_net_only=colour_only_per_ray(args.workload, args.precision),
                                                 note="value counts the reference's 256 network evaluations per ray; the fine pass's 64 coarse depths reuse the coarse pass's "
                                                      "hash features / outputs (identical results), so the kernels process fewer"),
            "rays_per_s": value / UNITS_PER_RAY, "s_per_frame": elapsed / args.steps / nframes * (world if args.scaling == "weak" else 1),
            # host side of the sharded step on rank 0: wall time spent INSIDE Render (one nrf_render_rows call per tile: ~25 asynchronous launches, no
            # synchronisation) -- the unsharded work that bounds strong scaling once a tile's kernels get short
            "host_ms_per_tile": host_s / args.steps / nframes * 1e3, "tile_rows": shard.rows,
            "roofline": roof, "kernel_ms": prof,
        }
        if other is not None:
            line["also"] = [other]
        if collective_check is not None:
            line["collective_check"] = collective_check
        if cpu is not None:
            line["cpu_baseline"] = cpu
        # parity of what was just timed (cpu_baseline leg, checker use of oracle/): GPU render vs the CPU oracle on identical weights/pose
        try:
            line["psnr_vs_oracle_db"] = quality_check(sc, renderer, rp, K, poses[0], args)
        except Exception as e:
            line["psnr_vs_oracle_db"] = f"unavailable: {e}"
        if not args.no_parity:
            try:
                line["parity_full_frame_vs_f32"] = full_frame_parity(sc, renderer, rp, K, poses[0], args, scene, L)
            except Exception as e:
                line["parity_full_frame_vs_f32"] = f"unavailable: {e}"
        assert frames.shape[0] == nframes and bool(torch.isfinite(frames).all())
        # the gathered frame of the first pose, hashed: equal strings at different N (or launchers) = the sharded render is the single-GPU render bit for bit
        import hashlib
        line["frame_sha256"] = hashlib.sha256(frames[0].reshape(H, W, 3).contiguous().cpu().numpy().tobytes()).hexdigest()
        if world == 1 and not use_dist and not args.no_also:
            line["also"] = secondary_measurements(args, scene, L, K, poses[0], sc)
    if rank == 0:
        # The contract is ONE JSON line.  RCCL prints a version banner through C stdio when the process exits, and tearing the process group down is a collective
        # that a slow or already-gone peer can stall: drain what is buffered, print the line, flush -- and in the distributed case every rank then leaves without
        # running tear-down or exit-time printers (everything that was to be measured and checked is in the line).
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    if use_dist:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


def timed_frames(L, render, frames, warm=2, lanes_hook=None):
    """warm untimed + `frames` timed calls of render() -> (seconds per frame, per-kernel HIP-event ms per frame and launches per frame)."""
    import ctypes as C
    import torch
    for _ in range(warm):
        render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        out = render()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    # the per-kernel times come from a second, single-lane pass (nrf_set_render_lanes(1): a kernel has the GPU to itself, its launch time is its own); the frame time
    # above is the default two-lane Chunk loop's
    n = len(L.NRF_PROF_NAMES)
    ms = (C.c_double * n)(); cnt = (C.c_int64 * n)()
    single = os.environ.get("NRF_RENDER_LANES", "2") != "1"
    if single:
        L.lib().nrf_set_render_lanes(1)
        if lanes_hook:
            lanes_hook(1)                        # a host whose own Chunk loop has lanes (LeRFRenderer)
    frames = max(2, min(frames, 4))
    render()
    torch.cuda.synchronize()
    L.lib().nrf_profile_enable(1)
    L.lib().nrf_profile_read(ms, cnt, 1)
    for _ in range(frames):
        out = render()
    torch.cuda.synchronize()
    L.lib().nrf_profile_read(ms, cnt, 1)
    if single:
        L.lib().nrf_set_render_lanes(2)
        if lanes_hook:
            lanes_hook(2)
    L.lib().nrf_profile_enable(0)
    return dt, {nm: dict(ms_per_frame=ms[i] / frames, launches_per_frame=cnt[i] / frames) for i, nm in enumerate(L.NRF_PROF_NAMES)}, out


def mfma_roofline(kernel, units, flop_per_unit, seconds, launches, issued_flop_per_unit=None, peak=MFMA_F16_PEAK, note=None):
    r = dict(bound="mfma", kernel=kernel, achieved=units * flop_per_unit / max(seconds, 1e-12) / 1e12, peak=peak / 1e12, unit="TFLOP/s",
             frac=units * flop_per_unit / max(seconds, 1e-12) / peak, launches=launches, avg_launch_ms=seconds * 1e3 / max(launches, 1),
             units_per_launch=units / max(launches, 1), flop_per_unit=flop_per_unit)
    if issued_flop_per_unit:
        r["mfma_issued_frac"] = units * issued_flop_per_unit / max(seconds, 1e-12) / peak
        r["mfma_issued_vs_sustained_gemm"] = units * issued_flop_per_unit / max(seconds, 1e-12) / MFMA_F16_SUSTAINED_GEMM
    if note:
        r["note"] = note
    return r


def secondary_measurements(args, scene, L, K, c2w, sc_main, steps=10):
    """Extra timings at N = 1 (not part of `value`): the other matrix-core precision of this workload and the other BASELINE workloads -- each with its own
    per-kernel HIP-event times, the roofline of its dominant kernel (algorithmic flops of the network as written x the evaluations the kernel executed) and its
    render-vs-oracle quality on a 256-ray sample."""
    import torch
    out = []
    # (workload, precision, coarse pass); classic split precision is timed in both coarse modes: "exact" (NRF_COARSE_AUTO: density branch in exact fp32 on the
    # matrix cores, the fp32 path's sample set bit for bit) and "full" (NRF_COARSE_FULL: whole network in split arithmetic, outputs reused: 99.9 % of pixels within 1e-4)
    todo = [("hash", "f16" if args.precision != "f16" else "f16x3", None), ("classic", "f16x3", "exact"), ("classic", "f16x3", "full"), ("classic", "f16", None)] if args.workload == "hash" else \
           [("classic", "f16x3", "full"), ("classic", "f16" if args.precision != "f16" else "f16x3", None), ("hash", "f16x3", None)]
    scenes = {args.workload: sc_main}
    for wl, pname, coarse in todo:
        try:
            if wl not in scenes:
                scenes[wl] = scene.make_hash_scene(mode=args.hash_mode) if wl == "hash" else scene.make_classic_scene()
            sc = scenes[wl]
            prec = {"f16": L.NRF_PREC_F16_MFMA, "f16x3": L.NRF_PREC_F16_SPLIT, "f32": L.NRF_PREC_F32}[pname]
            rp = scene.lego_render_params(sc["bbox"], NS, NI, 131072 if wl == "hash" else 8192, prec)
            if coarse == "full":
                rp.CoarseMode = L.NRF_COARSE_FULL
            n_fr = steps if not (wl == "classic" and pname == "f16x3") else max(3, steps // 2)
            dt, kms, _ = timed_frames(L, lambda: sc["renderer"].Render(H, W, K, rp, c2w=c2w), n_fr)
            a2 = argparse.Namespace(**{**vars(args), "workload": wl, "precision": pname, "coarse_full": coarse == "full"})
            ex_hash, ex_mlp, ex_sigma = executed_per_ray(wl, pname, args.hash_mode, coarse_full=(coarse != "exact") if wl == "classic" else False)
            rec = dict(workload="hashnerf_lego800_64+128" if wl == "hash" else "classic_nerf_lego800_64+128", baseline_config=2 if wl == "hash" else 1,
                       precision=pname, value=H * W * UNITS_PER_RAY / dt, unit="ray-samples/s", ms_per_step=dt * 1e3, steps=n_fr, kernel_ms=kms,
                       **({"coarse_pass": "density branch in exact fp32 on the matrix cores + colour branch on the exact h8 (sigma_nerf_f32.hip): the fp32 path's sample set, outputs reused by the fine pass" if coarse == "exact"
                           else "whole network in the timed arithmetic, outputs reused by the fine pass (NRF_COARSE_FULL)"} if coarse else {}),
                       executed_evaluations_per_ray=dict(hash_encode=ex_hash, fused_mlp=ex_mlp, sigma_only=ex_sigma, colour_net_only=colour_only_per_ray(wl, pname)))
            mk = kms["mlp"]
            if wl == "classic":
                rec["roofline"] = mfma_roofline("mlp_nerf" + ("_split" if pname == "f16x3" else ""), H * W * ex_mlp, NERF_FLOP_PER_UNIT, mk["ms_per_frame"] * 1e-3,
                                                mk["launches_per_frame"], issued_flop_per_unit=(3.0 if pname == "f16x3" else 1.0) * 1058 * 32768 / 32,
                                                note="algorithmic 1 186 816 flop of NeRFImpl::forward as written x the network evaluations the kernel executed "
                                                     "(the fine pass's 64 coarse depths take the coarse pass's outputs); issued: 1 058 matrix instructions per 32 points"
                                                     + (" x 3 products (hi + lo operand pairs)" if pname == "f16x3" else ""))
                sk = kms["sigma"]
                if sk["launches_per_frame"]:
                    rec["roofline"]["sigma_exact"] = mfma_roofline("sigma_nerf_f32 (coarse pass: density branch in exact fp32, v_mfma_f32_32x32x2_f32; + the colour branch in split fp16, 2 % of its matrix time)", H * W * ex_sigma, NERF_SIGMA_FLOP_PER_UNIT,
                                                                   sk["ms_per_frame"] * 1e-3, sk["launches_per_frame"], peak=F32_PEAK)
            else:
                rec["roofline"] = mfma_roofline("mlp_small", H * W * ex_mlp, SMALL_FLOP_PER_UNIT, mk["ms_per_frame"] * 1e-3, mk["launches_per_frame"],
                                                issued_flop_per_unit=SMALL_MFMA_FLOP_PER_UNIT.get(pname))
                ck = kms["mlp_colour"]
                if ck["launches_per_frame"]:
                    rec["roofline"]["colour_only"] = mfma_roofline("mlp_small, colour net alone", H * W * colour_only_per_ray(wl, pname), SMALL_COLOUR_FLOP_PER_UNIT, ck["ms_per_frame"] * 1e-3,
                                                                   ck["launches_per_frame"], issued_flop_per_unit=SMALL_COLOUR_MFMA_FLOP_PER_UNIT)
                hk = kms["hash"]
                rec["roofline"]["hash"] = dict(bound="hbm", kernel="hash_encode", unit="GB/s", peak=HBM_PEAK / 1e9,
                                               achieved=H * W * ex_hash * HASH_BYTES_PER_UNIT / max(hk["ms_per_frame"] * 1e-3, 1e-12) / 1e9,
                                               frac=H * W * ex_hash * HASH_BYTES_PER_UNIT / max(hk["ms_per_frame"] * 1e-3, 1e-12) / HBM_PEAK)
            rec["psnr_vs_oracle_db"] = quality_check(sc, sc["renderer"], rp, K, c2w, a2)
            out.append(rec)
        except Exception as e:
            out.append(dict(workload=wl, precision=pname, error=str(e)))
    if args.workload == "hash" and args.hash_mode == "cu":
        # the same configuration on the LibTorch HashEmbedder + SHEncoder (the encoders whose reference implementation runs on the CPU and
        # pins the oracle) -- fp32 table, hi + lo fp16 feature planes
        try:
            sc = scene.make_hash_scene(mode="ngp")
            rp = scene.lego_render_params(sc["bbox"], NS, NI, 131072, L.NRF_PREC_F16_SPLIT)
            dt, kms, _ = timed_frames(L, lambda: sc["renderer"].Render(H, W, K, rp, c2w=c2w), steps)
            a2 = argparse.Namespace(**{**vars(args), "workload": "hash", "precision": "f16x3", "hash_mode": "ngp"})
            out.append(dict(workload="hashnerf_lego800_64+128", baseline_config=2, encoder="HashEmbedder + SHEncoder (LibTorch twin)", precision="f16x3",
                            value=H * W * UNITS_PER_RAY / dt, unit="ray-samples/s", ms_per_step=dt * 1e3, steps=steps, kernel_ms=kms,
                            psnr_vs_oracle_db=quality_check(sc, sc["renderer"], rp, K, c2w, a2)))
            del sc
            torch.cuda.empty_cache()
        except Exception as e:
            out.append(dict(workload="hashnerf (HashEmbedder twin)", error=str(e)))
    try:
        out.append(train_step_measurement(args, scene, L))
    except Exception as e:
        out.append(dict(workload="hashnerf_train_step", error=str(e)))
    for lp in (L.NRF_PREC_F16_SPLIT, L.NRF_PREC_F16_MFMA):
        try:
            out.append(lerf_measurement(scene, L, K, c2w, lp))
        except Exception as e:
            out.append(dict(workload="lerf_lego800_64+128", error=str(e)))
    return out


def lerf_oracle_check(sc, res, nrays=256):
    """`nrays` rays of the LeRF frame end to end through the CPU oracle's fp32 stage path (CuHash F = 8 encode -> LeRFImpl::forward -> RawToLEOutputs weights ->
    SamplePDF -> fine pass -> RenderCLIPEmbedding): sample set, weights and rendered embedding of the GPU pass against it.  Checker use of oracle/ only."""
    import torch
    from oracle import capi as O
    from nerfpp_amd import scene
    acc = res.Outputs.AccMapLE.cpu().numpy()
    hit = np.nonzero(acc > 1e-2)[0]
    idx = hit[::max(1, hit.size // nrays)][:nrays]
    rays = res.Extras["rays_flat"].cpu().numpy()[idx]
    Lv, F, T = 16, 8, 19
    ls = ((1 << T) >> 4) << 4
    tab16 = O.f32_to_f16(sc["table"])
    mul = O.hash_cu_scales(Lv, 16, 1024)

    def net(pts):
        e_, keep = O.hash_cu(pts.reshape(-1, 3), tab16, sc["primes"], np.arange(Lv, dtype=np.int32) * ls, np.full(Lv, ls, np.int32), np.zeros((Lv, 3), np.float32), sc["bbox"], mul, Lv, F)
        o = O.lerf(sc["blob"], e_)
        o[~keep, -1] = 0
        return o.reshape(pts.shape[0], pts.shape[1], -1)
    zc = O.z_vals(rays[:, 6], rays[:, 7], O.linspace(0, 1, NS))
    wc = O.raw2weights(net(O.points(rays[:, :3], rays[:, 3:6], zc)), 768, zc, rays[:, 3:6])["weights"]
    samples, _, _ = O.sample_pdf(O.z_mid(zc), wc[:, 1:-1], O.linspace(0, 1, NI))
    zf = O.merge_sorted(zc, samples)
    rawf = net(O.points(rays[:, :3], rays[:, 3:6], zf))
    fin = O.raw2weights(rawf, 768, zf, rays[:, 3:6])
    ref = O.render_clip_embedding(rawf, 768, fin["weights"])
    zg = res.Extras["z_fine"].cpu().numpy()[idx]
    wg = res.Outputs.WeightsLE.cpu().numpy()[idx]
    eg = res.Outputs.RenderedLangEmbedding.cpu().numpy()[idx]
    cos = (eg * ref).sum(1)
    same = (zg == zf).all(1)
    return dict(rays=int(idx.size), embedding_max_abs_err=float(np.abs(eg - ref).max()), embedding_rms_err=float(np.sqrt(((eg - ref).astype(np.float64) ** 2).mean())),
                fine_sample_set_bit_identical_rays=float(same.mean()), weights_max_abs_err_over_max=float(np.abs(wg - fin["weights"])[same].max() / fin["weights"].max()) if same.any() else None,
                embedding_cos_min=float(cos.min()), embedding_cos_min_same_samples=float(cos[same].min()) if same.any() else None, embedding_cos_median=float(np.median(cos)),
                against="CPU oracle, fp32 stage path end to end (its own coarse pass and fine sample set)")


def lerf_measurement(scene, L, K, c2w, precision, repeats=10):
    """BASELINE config 5: the LeRF language-embedding render pass (CuHashEmbedder L16 F8 T2^19 16..1024 + LeRF 2x256 -> 768, main.cpp:203-213)
    on the WHOLE 800x800 frame, 64+128 samples: warm-up, then `repeats` timed frames with per-kernel HIP-event times, rooflines and an oracle check of 256 rays."""
    import torch
    from nerfpp_amd import renderer as R
    sc = scene.make_lerf_scene()
    p = R.NeRFRenderParams(NSamples=NS, NImportance=NI, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True,
                           BoundingBox=sc["bbox"])
    r = sc["renderer"]
    r.set_precision(precision)
    dt, kms, res = timed_frames(L, lambda: r.Render(H, W, K, p, c2w=c2w), repeats, warm=1, lanes_hook=lambda k: setattr(r, "lanes", k))
    n = H * W
    emb = res.Outputs.RenderedLangEmbedding
    hit = res.Outputs.AccMapLE > 1e-2
    nrm = emb[hit].norm(dim=1)
    split = r.precision_name == "f16x3"
    exact = bool(split and r._exact_coarse_on())
    rec = dict(workload="lerf_lego800_64+128", baseline_config=5, rays=n, value=n * UNITS_PER_RAY / dt, unit="ray-samples/s", s_per_frame=dt, frames_timed=repeats, kernel_ms=kms,
               fused_matrix_core_path=bool(r.fused), finite=bool(torch.isfinite(emb).all()),
               rays_with_language_density=int(hit.sum()), embedding_norm_min_max=[float(nrm.min()), float(nrm.max())] if int(hit.sum()) else None,
               level_major_features=bool(getattr(r, "level_major", False)), precision=getattr(r, "precision_name", "f16"),
               coarse_pass="sigma_le in exact fp32 on the matrix cores (sigma_lerf_f32.hip): the fp32 path's fine sample set" if exact else "the timed arithmetic",
               executed_evaluations_per_ray=dict(hash_encode=NS + NI, density_net=NS + NI, embedding_net=NS + NI,
                                                 note="every sample point is encoded once and its density net evaluated once (the fine pass's 64 coarse depths reuse the coarse pass's columns)"),
               arithmetic=("split-f16 MFMA (hi + lo operand pairs, three products, fp32 accumulate: fp32-grade)" if split else "fp16 MFMA (fp32 accumulate)") +
                          " LeRF head fused with the render pass; CuHash F=8 features level-major fp16 (256 B per sample point), "
                          "read by the kernels as operand fragments; embedding norm via the Gram matrix of the bias-free output layer, which is applied once per ray to the weighted sum of its inputs")
    # rooflines: algorithmic flops of LeRFImpl::forward as written (557 568 per sample, the 256 -> 768 layer per SAMPLE) x the 192 evaluations per ray the passes execute
    mk, sk, hk = kms["mlp"], kms["sigma"], kms["hash"]
    units = n * (NS + NI)
    t_mlp = (mk["ms_per_frame"] + sk["ms_per_frame"]) * 1e-3
    issued = None
    if split:
        new_pts, all_pts = n * NI, n * (NS + NI)
        issued_f16 = ((new_pts if exact else all_pts) * LERF_SPLIT_MFMA_SIGMA + all_pts * LERF_SPLIT_MFMA_EMBED) * 32768 / 32 + n * 24 * 16 * 3 * 32768 / 32
        issued = issued_f16 / units
    rec["roofline"] = mfma_roofline("lerf passes (density net + embedding net + per-ray output layer" + ("; coarse density net: exact-fp32 kernel, in `sigma_exact`" if exact else "") + ")",
                                    units, LERF_FLOP_PER_UNIT, t_mlp, mk["launches_per_frame"] + sk["launches_per_frame"], issued_flop_per_unit=issued,
                                    note="achieved / frac price the ALGORITHMIC 557 568 flop per sample of LeRFImpl::forward as written over the time of ALL LeRF network kernels; the kernels "
                                         "execute far fewer (Gram-matrix norm, output layer once per ray): mfma_issued_frac is the fp16 matrix pipe's own share")
    if exact and sk["launches_per_frame"]:
        rec["roofline"]["sigma_exact"] = mfma_roofline("lerf_sigma_f32 (coarse pass: density net in exact fp32, v_mfma_f32_32x32x2_f32)", n * NS, LERF_SIGMA_FLOP_PER_UNIT, sk["ms_per_frame"] * 1e-3,
                                                       sk["launches_per_frame"], peak=F32_PEAK)
    if hk["launches_per_frame"]:
        b = units * LERF_HASH_BYTES_PER_UNIT / max(hk["ms_per_frame"] * 1e-3, 1e-12)
        rec["roofline"]["hash"] = dict(bound="hbm", kernel="hash_encode F=8 (k_hash_cu, level-major fp16 out)", unit="GB/s", achieved=b / 1e9, peak=HBM_PEAK / 1e9, frac=b / HBM_PEAK,
                                       frac_of_infinity_cache_gather_rate=b / GATHER_PEAK, bytes_per_unit=LERF_HASH_BYTES_PER_UNIT, units_per_frame=units,
                                       note="the 134 MB hashed table is Infinity-Cache resident: the gather path's ceiling for such tables is 8.6 TB/s (MI355X_MICROARCH.md)")
    try:
        rec["oracle_check"] = lerf_oracle_check(sc, res)
    except Exception as e:
        rec["oracle_check"] = f"unavailable: {e}"
    return rec


def train_step_measurement(args, scene, L, n_rand=16384, steps=5, mlp_backward="f16"):
    """one optimisation step of NeRFExecutor::Train (render the ray batch, huber loss, backward of the fine pass,
    Adam) on the HashNeRF configuration, N_rand = 32*32*16 rays per step as in the reference's main.cpp:232, next to the reference's own
    LibTorch CPU step (oracle/_ref/ref_driver bench_train) on a bounded ray batch."""
    import torch
    from nerfpp_amd import renderer as R
    from nerfpp_amd.train import Trainer
    sc = scene.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
    K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = R.GetRays(H, W, K, c2w)
    idx = torch.arange(0, n_rand, device="cuda") * (H * W // n_rand)
    o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
    tgt = torch.rand((n_rand, 3), device="cuda")
    tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward=mlp_backward, hash_backward="binned" if mlp_backward == "f16" else "f32")
    rp = R.NeRFRenderParams(NSamples=NS, NImportance=NI, Chunk=n_rand, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True,
                            BoundingBox=scene.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
    for _ in range(2):
        tr.step(o, d, tgt, rp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for _ in range(steps):
        lm, _ = tr.step(o, d, tgt, rp)
        losses.append(lm)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    rec = dict(workload="hashnerf_train_step", rays_per_step=n_rand, samples="64+128", ms_per_step=dt * 1e3, rays_per_s=n_rand / dt,
               value=n_rand * UNITS_PER_RAY / dt, unit="ray-samples/s", steps=steps, loss_first_last=[float(losses[0][0]), float(losses[-1][0])],
               arithmetic="render: split-f16 MFMA; NeRFSmall backward: " + ("one fused matrix-core kernel, fp16 operands / fp32 accumulation / device-side loss scale" if mlp_backward == "f16"
                                                                            else "fp32 layer-wise kernels") +
                          "; hash backward: ray-coherent fp32 pre-sum, then " + ("fixed-point records binned by table range and summed in LDS (no atomics to memory; equals the packed-atomic path bit for bit)" if mlp_backward == "f16" else "one float atomic per feature") +
                          "; Adam fp32")
    if mlp_backward == "f16":
        try:
            r32 = train_step_measurement(argparse.Namespace(**{**vars(args), "no_cpu_baseline": True}), scene, L, n_rand, steps, "f32")
            rec["fp32_backward"] = dict(ms_per_step=r32["ms_per_step"], rays_per_s=r32["rays_per_s"], loss_first_last=r32["loss_first_last"])
        except Exception as e:
            rec["fp32_backward"] = f"unavailable: {e}"
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(drv) and not args.no_cpu_baseline:
        try:
            outp = subprocess.run([drv, "bench_train", "1024", str(NS), str(NI), "4096", "1"], capture_output=True, text=True, timeout=600)
            r = json.loads(outp.stdout.strip().splitlines()[-1])
            rec["cpu_reference"] = dict(rays_per_s=r["rays_per_s"], value=r["units_per_s"], unit="ray-samples/s", cores=r["threads"], kind="reference",
                                        sample=f"{r['rays']} rays per step, LibTorch CPU HashEmbedder+SHEncoder+NeRFSmall forward+backward+Adam, {r['seconds']:.1f} s per step")
        except Exception as e:
            rec["cpu_reference"] = f"unavailable: {e}"
    return rec


# source files whose contents decide a kernel's HBM traffic: the PMC summary records their hashes, and a summary taken from other sources is not reported
PMC_KERNEL_SOURCES = {
    "hash_encode (k_hash_cu_lm)": ["hash_fast.hip", "hash_fast.h", "encode.h"],
    "mlp_small (k_mlp_small_mfma)": ["mlp_small_mfma.hip"],
    "sigma_small_f32 (k_sigma_small_f32)": ["sigma_small_f32.hip"],
    "mlp_nerf_split (k_mlp_nerf_split)": ["mlp_nerf_split_mfma.hip", "mlp_nerf_net.h"],
    "mlp_nerf (k_mlp_nerf_mfma)": ["mlp_nerf_mfma.hip", "mlp_nerf_net.h"],
}


def kernel_source_hash(kernel):
    import hashlib
    h = hashlib.sha256()
    for f in PMC_KERNEL_SOURCES.get(kernel, []):
        with open(os.path.join(ROOT, "nerfpp_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, units_per_launch, meta_key=None):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/pmc_latest.json, written from separate --pmc FETCH_SIZE / WRITE_SIZE
    runs of this same bench: bytes per point the kernel processed), times this run's points per launch.  The summary is stamped with the commit it was taken at
    and with a hash of each kernel's sources: if the kernel's source differs from what was profiled, NO traffic is reported (None) and the reason is given."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None, None
    try:
        d = json.load(open(path))
        meta = d["_meta"]
        stamp = "PMC pass at commit " + str(meta.get("commit", "unrecorded (round 2)"))
        want = (meta.get("kernel_source_sha256_16") or {}).get(kernel)
        if want is None or want != kernel_source_hash(kernel):
            return None, f"not reported: the kernel's sources changed since the {stamp} (profiles/pmc_latest.json); re-run tools/gpu_pmc_round.sh"
        return d[kernel]["hbm_bytes_per_point"] * units_per_launch, ("profiles/pmc_latest.json, " + stamp + ", kernel sources unchanged since: (2*FETCH_SIZE + WRITE_SIZE) KB per dispatch "
                                                                    "(gfx950 x2 read correction) summed over `bench.py --steps 1 --warmup 1` and divided by the points processed; " + meta["source"])
    except Exception:
        return None, None


def pmc_mfma_busy(kernel, precision):
    """Share of GPU-active cycles in which the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD), from the committed PMC
    passes; with the issued fraction at the nominal 2.4 GHz it gives the clock the chip held: clock = 2.4 GHz * issued_frac / busy_frac."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        return float(d[kernel]["mfma_busy_frac_of_active_cycles"][precision])
    except Exception:
        return None


def full_frame_parity(sc, renderer, rp, K, c2w, args, scene, L):
    """The whole frame just timed against this library's own NRF_PREC_F32 mode (which equals the CPU oracle bit for bit -- tests/ and the 256-ray sample
    below) on identical weights and pose: every pixel value of the 800x800 frame (a 100-row band for the classic 8x256 network, whose fp32 path takes seconds per frame)."""
    import copy
    import torch
    if args.precision == "f32":
        return "the timed mode IS the parity mode"
    rows = H if args.workload == "hash" else 100
    row0 = (H - rows) // 2
    a = renderer.Render(H, W, K, rp, c2w=c2w, row0=row0, rows=rows)
    rp32 = copy.copy(rp); rp32.Precision = L.NRF_PREC_F32; rp32.Chunk = 32768 if args.workload == "hash" else 8192
    b = renderer.Render(H, W, K, rp32, c2w=c2w, row0=row0, rows=rows)
    d = (a.Outputs.RGBMap - b.Outputs.RGBMap).abs()
    mse = float((d.double() ** 2).mean())
    return dict(pixels=int(rows * W), max_abs_err=float(d.max()), median_abs_err=float(d.median()), frac_within_1e4=float((d < 1e-4).float().mean()),
                psnr=(float("inf") if mse == 0 else -10.0 * float(np.log10(mse))), against="NRF_PREC_F32 (bit-exact with the CPU oracle) on the same frame")


def quality_check(sc, renderer, rp, K, c2w, args, nrays=256):
    """Second half of the cpu_baseline leg (outside every timed region): the CPU oracle renders 256 rays of the frame with the same weights
    and the GPU pixels are compared with it -- oracle/ is used as the checker only, never as part of what is measured or shipped."""
    import torch
    from oracle import capi as O
    from nerfpp_amd import scene
    res = renderer.Render(H, W, K, rp, c2w=c2w, row0=H // 2, rows=1)
    rays = res.Extras["rays_flat"].cpu().numpy()[::W // nrays][:nrays]
    rgb = res.Outputs.RGBMap.cpu().numpy().reshape(-1, 3)[::W // nrays][:nrays]
    if args.workload == "hash":
        cfg = sc["cfg"]
        if sc["mode"] == "cu":
            ls = ((1 << cfg["log2_t"]) >> 4) << 4
            Lv = cfg["n_levels"]
            model = O.Model(2, sc["mlp_blob"], bbox=sc["bbox"], table_f16=O.f32_to_f16(sc["table"]), primes=sc["primes"],
                            local_idx=np.arange(Lv, dtype=np.int32) * ls, local_size=np.full(Lv, ls, np.int32), bias=np.zeros((Lv, 3), np.float32),
                            mul=O.hash_cu_scales(Lv, cfg["base"], cfg["finest"]))
        else:
            model = O.Model(0, sc["mlp_blob"], bbox=sc["bbox"], table_f32=sc["table"])
    else:
        model = O.Model(1, sc["mlp_blob"], bbox=sc["bbox"])
    ref = O.render_rays(model, rays, NS, NI, O.linspace(0, 1, NS), O.linspace(0, 1, NI), white_bkgr=True)
    d = np.abs(rgb - ref["rgb"])
    return dict(psnr=scene.psnr(rgb, ref["rgb"]), max_abs_err=float(d.max()), median_abs_err=float(np.median(d)), frac_within_1e4=float((d < 1e-4).mean()),
                rays=int(rays.shape[0]))


if __name__ == "__main__":
    main()
