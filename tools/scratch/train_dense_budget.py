"""Training step time vs the hash grid's baked-pyramid budget (re-baked at every table upload).  usage (GPU box): python tools/scratch/train_dense_budget.py"""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
from nerfpp_amd.train import Trainer
H = W = 800; N = 16384
sc = S.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(H, W, K, c2w)
idx = torch.arange(0, N, device="cuda") * (H * W // N)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
tgt = torch.rand((N, 3), device="cuda")
rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=N, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward="f16", hash_backward="binned")
for rep in range(2):
    for mb in (0, 4, 16, 64, 256, 1024):
        sc["embedder"].set_dense_budget(mb << 20)
        for _ in range(2): tr.step(o, d, tgt, rp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): lm, _ = tr.step(o, d, tgt, rp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
        print(f"dense budget {mb:5d} MB: step {dt*1e3:6.2f} ms  loss {float(lm[0]):.5f}", flush=True)
