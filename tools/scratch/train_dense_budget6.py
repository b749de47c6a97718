"""Hash training step against the dense-image budget kept while the table moves (Trainer(train_dense_budget=...)): the re-bake after every step costs, the baked levels' 2-load lookups save."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
from nerfpp_amd.train import Trainer
N = 16384
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(800, 800, K, c2w)
idx = torch.arange(0, N, device="cuda") * (640000 // N)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
tgt = torch.rand((N, 3), device="cuda")
rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=N, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
for mb in (256, 0, 16, 64, 128, 256, 1024):
    sc = S.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
    tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward="f16", hash_backward="binned", train_dense_budget=mb << 20)
    for _ in range(3): tr.step(o, d, tgt, rp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.step(o, d, tgt, rp)
    torch.cuda.synchronize()
    print("train_dense_budget %5d MB: %.3f ms per step" % (mb, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    tr.close() if hasattr(tr, "close") else None
