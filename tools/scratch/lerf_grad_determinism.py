"""Is the LeRF backward's gradient the same from call to call?  One render, the backward N times on it (the language-grid scatter uses float atomics: last-bit
differences in g_table are expected; the head's gradient goes through the deterministic GEMM kernels)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
from nerfpp_amd.train import LeRFTrainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
sc = S.make_lerf_scene()
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(800, 800, K, c2w)
idx = torch.arange(0, n, device="cuda") * (640000 // n)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
tgt = torch.nn.functional.normalize(torch.randn((n, 768), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)), dim=-1)
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=False, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"],
                       KeepIntermediates=True)
tr = LeRFTrainer(sc["renderer"], sc["table"], sc["blob"], learning_rate=5e-4)
res = tr.renderer.Render(0, 0, None, p, rays=(o, d, None))
gs = []
for i in range(6):
    tr.backward(res, tgt, p, None)
    torch.cuda.synchronize()
    gs.append((tr.g_blob.clone(), tr.g_table.clone()))
for i in range(1, 6):
    db = (gs[i][0] - gs[0][0]).abs().max() / gs[0][0].abs().max(); dt = (gs[i][1] - gs[0][1]).abs().max() / gs[0][1].abs().max()
    print("call %d vs call 0: head gradient max diff / max %.3e   table gradient %.3e   finite %s" % (i, float(db), float(dt), bool(torch.isfinite(gs[i][0]).all())))
tr.close()

# ... and does it depend on what the (uninitialised) workspace held?  The same backward with the workspace pre-filled with zeros and with NaNs.
tr = LeRFTrainer(sc["renderer"], sc["table"], sc["blob"], learning_rate=5e-4)
res = tr.renderer.Render(0, 0, None, p, rays=(o, d, None))
tr.backward(res, tgt, p, None)
out = {}
for name, val in (("zeros", 0.0), ("nan", float("nan")), ("big", 3.0e38)):
    tr._ws.view(torch.float32)[: tr._ws.numel() // 4].fill_(val)
    tr.backward(res, tgt, p, None)
    torch.cuda.synchronize()
    out[name] = (tr.g_blob.clone(), tr.g_table.clone())
for name in ("nan", "big"):
    print("workspace pre-filled with %s vs zeros: head gradient equal %s (finite %s), table gradient equal %s" % (
        name, bool(torch.equal(out[name][0], out["zeros"][0])), bool(torch.isfinite(out[name][0]).all()), bool(torch.equal(out[name][1], out["zeros"][1]))))
tr.close()
