"""Host time spent inside each library entry during training steps (ctypes calls wrapped by a timing proxy).  usage (GPU box): python tools/scratch/lib_call_times.py [hash|lerf|classic]"""
import os, sys, time, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene, renderer as R
from nerfpp_amd import train as T
acc = collections.defaultdict(lambda: [0.0, 0])
real = L.lib()
class Proxy:
    def __getattr__(self, name):
        f = getattr(real, name)
        if not name.startswith("nrf_"): return f
        def timed(*a, **k):
            t0 = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t0
            acc[name][0] += dt; acc[name][1] += 1
            return r
        return timed
proxy = Proxy()
L.lib = lambda: proxy
for mod in (R, T):
    if hasattr(mod, "L"): mod.L.lib = L.lib
which = sys.argv[1] if len(sys.argv) > 1 else "hash"
H = W = 800
K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(H, W, K, c2w)
if which == "hash":
    n = 16384; sc = scene.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
    tr = T.Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward="f16", hash_backward="binned")
    rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=n, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=scene.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
    tgt = torch.rand((n, 3), device="cuda")
else:
    n = 4096; sc = scene.make_classic_scene()
    tr = T.Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], None, sc["mlp_blob"], learning_rate=5e-4)
    rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=n, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=scene.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
    tgt = torch.rand((n, 3), device="cuda")
idx = torch.arange(0, n, device="cuda") * (H * W // n)
oo = o.reshape(-1, 3)[idx].contiguous(); dd = d.reshape(-1, 3)[idx].contiguous()
for _ in range(3): tr.step(oo, dd, tgt, rp)
torch.cuda.synchronize(); acc.clear()
N = 10; t0 = time.perf_counter()
for _ in range(N): tr.step(oo, dd, tgt, rp)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("%s: %d steps, host %.2f ms per step, with final sync %.2f ms per step" % (which, N, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
for name, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-44s %7.3f ms per step over %4.1f calls" % (name, t / N * 1e3, c / N))
tr.close()
