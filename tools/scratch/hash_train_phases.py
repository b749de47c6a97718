"""HashNeRF training step (hash grid + NeRFSmall, the bench's hashnerf_train_step): host time to enqueue a step against its time with a synchronisation, and the parameter push alone.
usage (GPU box): python tools/scratch/hash_train_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene, renderer as R
from nerfpp_amd.train import Trainer
H = W = 800; n_rand = 16384
sc = scene.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(H, W, K, c2w)
idx = torch.arange(0, n_rand, device="cuda") * (H * W // n_rand)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
tgt = torch.rand((n_rand, 3), device="cuda")
tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward="f16", hash_backward="binned")
rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=n_rand, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=scene.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
for _ in range(3): tr.step(o, d, tgt, rp)
torch.cuda.synchronize()
N = 10; host = 0.0; tot = 0.0
for _ in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.step(o, d, tgt, rp); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host += t1 - t0; tot += t2 - t0
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): tr.step(o, d, tgt, rp)
torch.cuda.synchronize(); back = (time.perf_counter() - t0) / N
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): tr._push_params()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("step: host enqueue %.2f ms, with sync %.2f ms, back to back %.2f ms; parameter push: host %.2f ms, with sync %.2f ms" % (host / N * 1e3, tot / N * 1e3, back * 1e3, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
tr.close()
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward="f16", hash_backward="binned")
    for _ in range(3): tr.step(o, d, tgt, rp)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): tr.step(o, d, tgt, rp)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
