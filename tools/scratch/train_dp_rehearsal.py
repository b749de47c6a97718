"""Data-parallel training step rehearsal (SURVEY 8f row N1 across ranks): 2 ranks share the box's ONE GPU over gloo, each back-propagates its own half of a
16 384-ray batch against the replicated model, GradSync agrees on the overflow flag and averages the gradients (67 MB table gradient + the MLP blob, bucketed),
both take the same Adam step.  Checks: parameters stay identical on the two ranks, and equal the single-rank step on the whole batch up to the summation order of
the gradient mean.  Not a scaling measurement (gloo stages through the host; one GPU serves both ranks): it exercises the code path the driver's N > 1 run needs.

    python tools/scratch/train_dp_rehearsal.py            # parent: starts the two ranks (no GPU call), prints rank 0's JSON line
"""
import json, os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def worker():
    import numpy as np, torch, torch.distributed as dist
    from nerfpp_amd import _lib as L, scene as S, renderer as R
    from nerfpp_amd.train import Trainer
    from nerfpp_amd.dist import GradSync
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H = W = 800; N = 16384
    sc = S.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
    K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = R.GetRays(H, W, K, c2w)
    idx = torch.arange(0, N, device="cuda") * (H * W // N)
    o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    tgt = torch.rand((N, 3), device="cuda", generator=g)
    half = N // world
    sl = slice(rank * half, (rank + 1) * half)
    rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=N, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True,
                            BoundingBox=S.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
    tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward="f16", hash_backward="binned",
                 grad_sync=GradSync())
    for _ in range(2):
        tr.step(o[sl], d[sl], tgt[sl], rp)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        lm, _ = tr.step(o[sl], d[sl], tgt[sl], rp)
    torch.cuda.synchronize(); dist.barrier()
    dt = (time.perf_counter() - t0) / steps
    # replicas identical?
    chk = torch.stack([tr.table.double().sum(), tr.blob.double().sum(), tr.table.double().abs().sum()]).cpu()
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    same = all(torch.equal(both[0], b) for b in both)
    if rank == 0:
        print(json.dumps(dict(workload="hashnerf_train_step_data_parallel_rehearsal", ranks=world, backend="gloo (both ranks on one GPU)", rays_per_rank=half,
                              ms_per_step=dt * 1e3, steps=steps, replicas_bit_identical=bool(same), skipped_steps=getattr(tr, "skipped_steps", 0),
                              loss=[float(x) for x in lm.cpu().numpy()])), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if "RANK" in os.environ:
        worker()
        sys.exit(0)
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = [p.wait(timeout=900) for p in procs]
    sys.exit(max(rc))
