"""The LeRF training loop's loss over N steps (Python mirror, the bench's scene and batch): run it twice and compare -- is the trajectory reproducible?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
from nerfpp_amd.train import LeRFTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 14
n = 16384
sc = S.make_lerf_scene()
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(800, 800, K, c2w)
idx = torch.arange(0, n, device="cuda") * (640000 // n)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
tgt = torch.nn.functional.normalize(torch.randn((n, 768), device="cuda", generator=torch.Generator(device="cuda").manual_seed(77)), dim=-1)
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=False, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
tr = LeRFTrainer(sc["renderer"], sc["table"], sc["blob"], learning_rate=5e-4)
ls = []
for i in range(steps):
    l, _ = tr.step(o, d, tgt, p)
    ls.append(round(float(l.item()), 6))
print("mode", L.lib().nrf_get_train_gemm(), ls)
tr.close()
