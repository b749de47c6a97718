import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S, renderer as R
from nerfpp_amd.train import Trainer
H = W = 800
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
sc = S.make_hash_scene(mode="cu", table_amp=1e-2, sigma_scale=4.0)
K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(H, W, K, c2w)
idx = torch.arange(0, N, device="cuda") * (H * W // N)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
tgt = torch.rand((N, 3), device="cuda")
MB = sys.argv[2] if len(sys.argv) > 2 else "f32"; HB = sys.argv[3] if len(sys.argv) > 3 else "f32"
tr = Trainer(sc["embedder"], sc["embeddirs"], sc["mlp"], sc["table"], sc["mlp_blob"], learning_rate=5e-4, mlp_backward=MB, hash_backward=HB)
for prec, name in ((L.NRF_PREC_F16_SPLIT, "f16x3"), (L.NRF_PREC_F32, "f32"))[:1 if "--fast-only" in sys.argv else 2]:
    rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=N, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True,
                            BoundingBox=S.LEGO_BBOX, Precision=prec)
    for _ in range(2): tr.step(o, d, tgt, rp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): lm, _ = tr.step(o, d, tgt, rp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(name, "step ms %.2f" % (dt * 1e3), "rays/s %.3e" % (N / dt), "units/s %.3e" % (N * 256 / dt), "loss", lm.cpu().numpy())
# phase timing of one step (f16x3 render)
rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=N, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX,
                        Precision=L.NRF_PREC_F16_SPLIT, ReturnRaw=True, KeepIntermediates=True)
import ctypes as C
def timed(f, label):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); print("  %-28s %.2f ms" % (label, (time.perf_counter() - t0) * 1e3)); return r
res = timed(lambda: tr.renderer.Render(0, 0, None, rp, rays=(o, d, None)), "render (both passes)")
timed(lambda: tr.backward(res, tgt, 192, False), "loss + backward")
def adam():
    for prm, g, m, v in ((tr.table, tr.g_table, tr.m_table, tr.v_table), (tr.blob, tr.g_blob, tr.m_blob, tr.v_blob)):
        L.check(L.lib().nrf_adam_step(C.c_void_p(prm.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()), C.c_int64(prm.numel()),
                                      C.c_float(5e-4), C.c_float(0.9), C.c_float(0.99), C.c_float(1e-15), 5, None))
timed(adam, "adam (16.8 M + 17.5 K params)")
timed(tr._push_params, "push params (fp16 table, repack)")
