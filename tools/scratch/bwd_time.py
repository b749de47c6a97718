"""time nrf_mlp_backward (fp32) vs nrf_mlp_backward_f16 on one training batch's worth of points"""
import sys, time, ctypes as C, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, modules as M
P = lambda t: C.c_void_p(t.data_ptr())
p = int(sys.argv[1]) if len(sys.argv) > 1 else 16384 * 192
rng = np.random.default_rng(1)
desc = L.MlpSmallDesc(32, 16, 3, 64, 15, 4, 64); lib = L.lib()
n_params = lib.nrf_mlp_small_param_count(C.byref(desc))
blob = (rng.standard_normal(n_params) * 0.18).astype(np.float32)
m = M.NeRFSmall(3, 64, 15, 4, 64, False, 3, 64, 32, 16, "model", params=blob)
x = (torch.rand((p, 48), device="cuda") * 2 - 1)
gr = torch.randn((p, 4), device="cuda") * 3e-6
for name, fn, wsfn in (("f32", lib.nrf_mlp_backward, lib.nrf_mlp_backward_workspace_bytes), ("f16", lib.nrf_mlp_backward_f16, lib.nrf_mlp_backward_f16_workspace_bytes)):
    nb = wsfn(m._m, C.c_int64(p)); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    g_blob = torch.zeros(n_params, device="cuda"); g_x = torch.zeros((p, 32), device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L.check(fn(m._m, P(x), P(gr), C.c_int64(p), P(g_blob), P(g_x), P(ws), C.c_size_t(nb), None)); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(name, "p", p, "ms %.2f" % (dt * 1e3), "ws MB %.0f" % (nb / 1e6), "Gpts/s %.2f" % (p / dt / 1e9))
