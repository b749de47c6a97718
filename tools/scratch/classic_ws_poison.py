"""The classic training step with the backward's workspace filled with NaN before each step: a backward that reads what it did not write shows (loss / parameters go NaN)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene, renderer as R
from nerfpp_amd.train import Trainer
H = W = 800
sc = scene.make_classic_scene()
K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(H, W, K, c2w)
idx = torch.arange(0, 4096, device="cuda") * (H * W // 4096)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
torch.manual_seed(5); tgt = torch.rand((4096, 3), device="cuda")
tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], None, sc["mlp_blob"], learning_rate=5e-4)
rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=4096, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=scene.LEGO_BBOX,
                        Precision=L.NRF_PREC_F16_SPLIT)
losses = []
for i in range(6):
    if getattr(tr, "_ws", None) is not None:
        tr._ws[: tr._ws.numel() // 4 * 4].view(torch.float32).fill_(float("nan"))
    l, _ = tr.step(o, d, tgt, rp)
    losses.append(float(l[0]))
    print("step", i, "loss", losses[-1], "g_blob finite", bool(torch.isfinite(tr.g_blob).all()), "blob finite", bool(torch.isfinite(tr.blob).all()), flush=True)
