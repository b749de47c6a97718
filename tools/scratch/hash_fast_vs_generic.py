"""fast-path (baked, level-major fp16) encode vs the generic kernel on many random points: count of differing features per level"""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, _lib as L
from nerfpp_amd.modules import _ptr, _stream
sc = S.make_hash_scene(mode="cu"); e = sc["embedder"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
g = torch.Generator(device="cuda"); g.manual_seed(3)
pts = (torch.rand((n, 3), device="cuda", generator=g) * 3.0 - 1.5).contiguous()
x = torch.empty((16, n, 2), device="cuda", dtype=torch.float16); k = torch.empty((n,), device="cuda", dtype=torch.uint8)
L.check(L.lib().nrf_hash_encode_lm_f16(e._h, _ptr(pts), C.c_int64(n), _ptr(x), _ptr(k), _stream()))
bad = torch.zeros(16, dtype=torch.int64)
for i in range(0, n, 1_000_000):
    emb, keep = e.forward(pts[i:i + 1_000_000])                 # generic kernel, fp32 rows [p, 32] (values are fp16 numbers)
    ref = emb.reshape(-1, 16, 2).permute(1, 0, 2).to(torch.float16)
    d = (ref != x[:, i:i + 1_000_000]).any(-1).sum(1).cpu()
    bad += d
print("points", n, "differing (point, level) pairs per level:", bad.tolist())
