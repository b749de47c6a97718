"""fast-path encode vs the generic kernel on the renderer's own sample points (rows of the bench frame, coarse depths)"""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, renderer as R, _lib as L
from nerfpp_amd.modules import _ptr, _stream
sc = S.make_hash_scene(mode="cu"); e = sc["embedder"]
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
rp = S.lego_render_params(sc["bbox"], chunk=32768, precision=L.NRF_PREC_F16_SPLIT, KeepIntermediates="depths")
res = sc["renderer"].Render(800, 800, K, rp, c2w=c2w, row0=300, rows=40)
rays = res.Extras["rays_flat"]; z = res.Extras["z_fine"]
pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3).contiguous()
n = pts.shape[0]
x = torch.empty((16, n, 2), device="cuda", dtype=torch.float16); k = torch.empty((n,), device="cuda", dtype=torch.uint8)
L.check(L.lib().nrf_hash_encode_lm_f16(e._h, _ptr(pts), C.c_int64(n), _ptr(x), _ptr(k), _stream()))
bad = torch.zeros(16, dtype=torch.int64)
for i in range(0, n, 1_000_000):
    emb, keep = e.forward(pts[i:i + 1_000_000])
    ref = emb.reshape(-1, 16, 2).permute(1, 0, 2).to(torch.float16)
    bad += (ref != x[:, i:i + 1_000_000]).any(-1).sum(1).cpu()
print("points", n, "differing per level:", bad.tolist())
