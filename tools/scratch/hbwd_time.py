"""time nrf_hash_backward_rays vs the packed fixed-point variant on a training batch (16384 rays x 192 samples) of the Lego-like scene"""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S, renderer as R
P = lambda t: C.c_void_p(t.data_ptr())
H = W = 800; N = 16384
sc = S.make_hash_scene(mode=sys.argv[1] if len(sys.argv) > 1 else "cu", table_amp=1e-2, sigma_scale=4.0)
K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
o, d, _ = R.GetRays(H, W, K, c2w)
idx = torch.arange(0, N, device="cuda") * (H * W // N)
o = o.reshape(-1, 3)[idx].contiguous(); d = d.reshape(-1, 3)[idx].contiguous()
rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=N, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX,
                        Precision=L.NRF_PREC_F16_SPLIT, ReturnRaw=True, KeepIntermediates=True)
res = sc["renderer"].Render(0, 0, None, rp, rays=(o, d, None))
rays = res.Extras["rays_flat"]; z = res.Extras["z_fine"]; n, s = z.shape
pts = torch.empty((n * s, 3), device="cuda"); lib = L.lib()
L.check(lib.nrf_points(P(rays), rays.shape[1], P(z), C.c_int64(n), s, P(pts), None))
g = torch.randn((n * s, 32), device="cuda") * 1e-5
e = sc["embedder"]; gt = torch.zeros(e.table_elems(), device="cuda"); gq = torch.zeros_like(gt)
nb = lib.nrf_hash_backward_packed_workspace_bytes(e._h); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
for name, f in (("float atomics", lambda: L.check(lib.nrf_hash_backward_rays(e._h, P(pts), C.c_int64(n), s, P(g), P(gt), None))),
                ("packed fixed-point", lambda: L.check(lib.nrf_hash_backward_rays_packed(e._h, P(pts), C.c_int64(n), s, P(g), P(gq), P(ws), C.c_size_t(nb), None)))):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-20s %.2f ms" % (name, dt * 1e3))
a, b = gq.cpu().numpy() / 3, gt.cpu().numpy() / 3
print("max |diff| / max |g_table|: %.3e   rms diff / rms: %.3e" % (np.abs(a - b).max() / np.abs(b).max(), np.sqrt(((a - b) ** 2).mean() / (b ** 2).mean())))
