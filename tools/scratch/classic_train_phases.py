"""Where the classic (8x256 NeRFImpl) training step's wall time goes: the parameter push (nrf_mlp_set_params: host re-pack of the weight images) against the rest.
usage (GPU box): python tools/scratch/classic_train_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene, renderer as R
from nerfpp_amd.train import Trainer
H = W = 800; n_rand = 4096
sc = scene.make_classic_scene()
K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(H, W, K, c2w)
idx = torch.arange(0, n_rand, device="cuda") * (H * W // n_rand)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
tgt = torch.rand((n_rand, 3), device="cuda")
tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], None, sc["mlp_blob"], learning_rate=5e-4)
rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=n_rand, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=scene.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
tr.step(o, d, tgt, rp); torch.cuda.synchronize()
N = 3
t0 = time.perf_counter()
for _ in range(N): tr.step(o, d, tgt, rp)
torch.cuda.synchronize(); whole = (time.perf_counter() - t0) / N
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): tr._push_params()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("whole step %.2f ms; parameter push: host %.2f ms, with sync %.2f ms" % (whole * 1e3, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
tr.close()
