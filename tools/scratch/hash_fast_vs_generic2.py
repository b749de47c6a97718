"""fast-path encode vs the generic kernel on boundary / lattice points"""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, _lib as L
from nerfpp_amd.modules import _ptr, _stream
sc = S.make_hash_scene(mode="cu"); e = sc["embedder"]
g = torch.Generator(device="cuda"); g.manual_seed(5)
n = 2_000_000
base = (torch.rand((n, 3), device="cuda", generator=g) * 3.0 - 1.5)
sets = {}
b = base.clone(); ax = torch.randint(0, 3, (n,), device="cuda", generator=g); sgn = torch.randint(0, 2, (n,), device="cuda", generator=g).float() * 3.0 - 1.5
b[torch.arange(n), ax] = sgn; sets["one coordinate on the box face"] = b
sets["outside the box"] = base * 1.3
lat = torch.round((base + 1.5) / 3.0 * 16.0) / 16.0 * 3.0 - 1.5; sets["level-0 lattice points"] = lat
lat2 = torch.round((base + 1.5) / 3.0 * 512.0) / 512.0 * 3.0 - 1.5; sets["level-15 lattice points"] = lat2
m = base.clone(); m[:, 0] = lat[:, 0]; sets["x on the level-0 lattice"] = m
for name, pts in sets.items():
    pts = pts.contiguous()
    x = torch.empty((16, n, 2), device="cuda", dtype=torch.float16); k = torch.empty((n,), device="cuda", dtype=torch.uint8)
    L.check(L.lib().nrf_hash_encode_lm_f16(e._h, _ptr(pts), C.c_int64(n), _ptr(x), _ptr(k), _stream()))
    emb, keep = e.forward(pts)
    ref = emb.reshape(-1, 16, 2).permute(1, 0, 2).to(torch.float16)
    d = (ref != x).any(-1).sum(1).cpu().tolist()
    print(name, "differing per level:", d, "keep differs:", int((keep.to(torch.uint8) != k).sum()))
