"""Where a LeRF training step's wall time goes: host time to ENQUEUE each phase and the phase's time with a synchronisation behind it (render pass, loss + backward, Adam +
parameter push).  usage (GPU box): python tools/scratch/lerf_train_phases.py"""
import os, sys, time, copy, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L, scene, renderer as R
from nerfpp_amd.train import LeRFTrainer, _ptr, _stream
H = W = 800; n_rand = 16384
sc = scene.make_lerf_scene()
K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(H, W, K, c2w)
idx = torch.arange(0, n_rand, device="cuda") * (H * W // n_rand)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
tgt = torch.nn.functional.normalize(torch.randn((n_rand, 768), device="cuda"), dim=-1)
p0 = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=False, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
tr = LeRFTrainer(sc["renderer"], sc["table"], sc["blob"], learning_rate=5e-4)
tr.step(o, d, tgt, p0); torch.cuda.synchronize()
acc = {}
def phase(name, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    a = acc.setdefault(name, [0.0, 0.0]); a[0] += t1 - t0; a[1] += t2 - t0
    return r
N = 3
for _ in range(N):
    p = copy.copy(p0); p.KeepIntermediates, p.ReturnWeights = True, True
    res = phase("render", lambda: tr.renderer.Render(0, 0, None, p, rays=(o, d, None)))
    loss = phase("loss + backward", lambda: tr.backward(res, tgt, p, None))
    tr.t += 1
    def adam():
        b1, b2 = tr.betas
        for prm, g, m, v in ((tr.table, tr.g_table, tr.m_table, tr.v_table), (tr.blob, tr.g_blob, tr.m_blob, tr.v_blob)):
            L.check(L.lib().nrf_adam_step(_ptr(prm), _ptr(g), _ptr(m), _ptr(v), C.c_int64(prm.numel()), C.c_float(tr.lr), C.c_float(b1), C.c_float(b2), C.c_float(tr.eps), tr.t, _stream()))
    phase("adam", adam)
    phase("push params (set_table, set_params)", tr._push_params)
for k, (h, t) in acc.items():
    print("%-40s host enqueue %7.2f ms   with sync %7.2f ms" % (k, h / N * 1e3, t / N * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): tr.step(o, d, tgt, p0)
torch.cuda.synchronize(); print("whole step, no syncs inside: %.2f ms" % ((time.perf_counter() - t0) / N * 1e3))
tr.close()
