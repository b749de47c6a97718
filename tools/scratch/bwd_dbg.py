import sys, ctypes as C, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, modules as M
P = lambda t: C.c_void_p(t.data_ptr())
nl, nlc, p = 3, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(1)
desc = L.MlpSmallDesc(32, 16, nl, 64, 15, nlc, 64)
lib = L.lib()
n_params = lib.nrf_mlp_small_param_count(C.byref(desc))
blob = (rng.standard_normal(n_params) * 0.18).astype(np.float32)
m = M.NeRFSmall(nl, 64, 15, nlc, 64, False, 3, 64, 32, 16, "model", params=blob)
xh = np.concatenate([rng.uniform(-1, 1, (p, 32)), rng.uniform(-1, 1, (p, 16))], 1).astype(np.float32)
if "--f16x" in sys.argv: xh = xh.astype(np.float16).astype(np.float32)
x = torch.from_numpy(xh).cuda()
gr_h = (rng.standard_normal((p, 4)) * 3e-6).astype(np.float32)
if "--rgbonly" in sys.argv: gr_h[:, 3] = 0
if "--sigonly" in sys.argv: gr_h[:, :3] = 0
gr = torch.from_numpy(gr_h).cuda()
out = {}
for name, fn, wsfn in (("f32", lib.nrf_mlp_backward, lib.nrf_mlp_backward_workspace_bytes), ("f16", lib.nrf_mlp_backward_f16, lib.nrf_mlp_backward_f16_workspace_bytes)):
    nb = wsfn(m._m, C.c_int64(p)); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    g_blob = torch.zeros(n_params, device="cuda"); g_x = torch.zeros((p, 32), device="cuda")
    L.check(fn(m._m, P(x), P(gr), C.c_int64(p), P(g_blob), P(g_x), P(ws), C.c_size_t(nb), None)); torch.cuda.synchronize()
    out[name] = (g_blob.cpu().numpy(), g_x.cpu().numpy())
gb32, gx32 = out["f32"]; gb16, gx16 = out["f16"]
dims = [(32, 64)] + [(64, 64)] * (nl - 2) + [(64, 16)] + [(31, 64)] + [(64, 64)] * (nlc - 2) + [(64, 3)]
off = 0
for li, (i, o) in enumerate(dims):
    a, b = gb16[off:off + i * o].reshape(o, i), gb32[off:off + i * o].reshape(o, i)
    sc = np.abs(b).max(); d = np.abs(a - b)
    print(li, (i, o), "max", d.max() / sc, "rms", np.sqrt((d ** 2).mean()) / sc, "corr", np.corrcoef(a.ravel(), b.ravel())[0, 1], "row-err", (d.max(1) / sc).round(3)[:8], "col-err", (d.max(0) / sc).round(3)[:8])
    off += i * o
sx = np.abs(gx32).max(); d = np.abs(gx16 - gx32)
print("gx max", d.max() / sx, "rms", np.sqrt((d ** 2).mean()) / sx, "worst rows", np.argsort(-d.max(1))[:8], "cols", (d.max(0) / sx).round(3))
