"""The classic training step (bench scene, 4 096 rays) for N steps: the loss of every step, under the library's default layer products and under NRF_TRAIN_GEMM modes
given on the command line (e.g. `classic_train_losses.py 16 -1 0`): do the split-precision chains train as the fp32 chain does?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene, renderer as R
from nerfpp_amd.train import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
modes = [int(a) for a in sys.argv[2:]] or [-1, 0]
H = W = 800
for mode in modes:
    L.lib().nrf_set_train_gemm(mode)
    torch.manual_seed(5)
    sc = scene.make_classic_scene()
    K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
    o, d, _ = R.GetRays(H, W, K, c2w)
    idx = torch.arange(0, 4096, device="cuda") * (H * W // 4096)
    o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
    tgt = torch.rand((4096, 3), device="cuda")
    tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], None, sc["mlp_blob"], learning_rate=5e-4)
    rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=4096, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=scene.LEGO_BBOX,
                            Precision=L.NRF_PREC_F16_SPLIT)
    losses = []
    for _ in range(steps):
        l, _ = tr.step(o, d, tgt, rp)
        losses.append(float(l[0]))
    print("train_gemm %d:" % mode, " ".join("%.5f" % x for x in losses), flush=True)
