"""diagnostic build (-DNRF_BWD_TRACE, NRF_LIB_PATH): where wave 0 of each workgroup of the training backward kernel (k_small_bwd) spends its cycles, per section"""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, modules as M
P = lambda t: C.c_void_p(t.data_ptr())
p = 16384 * 192
rng = np.random.default_rng(1)
desc = L.MlpSmallDesc(32, 16, 3, 64, 15, 4, 64); lib = L.lib()
n_params = lib.nrf_mlp_small_param_count(C.byref(desc))
blob = (rng.standard_normal(n_params) * 0.18).astype(np.float32)
m = M.NeRFSmall(3, 64, 15, 4, 64, False, 3, 64, 32, 16, "model", params=blob)
x = (torch.rand((p, 48), device="cuda") * 2 - 1); gr = torch.randn((p, 4), device="cuda") * 3e-6
lm = len(sys.argv) > 1 and sys.argv[1] == "lm"          # the trainer's input form: level-major fp16 features + per-ray fp16 direction rows
feats = (torch.rand((16, p, 2), device="cuda") * 2 - 1).half(); dirs = (torch.rand((p // 192, 16), device="cuda") * 2 - 1).half()
nb = lib.nrf_mlp_backward_f16_workspace_bytes(m._m, C.c_int64(p)); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
g_blob = torch.zeros(n_params, device="cuda"); g_x = torch.zeros((p, 32), device="cuda")
dbg = C.CDLL(os.environ["NRF_LIB_PATH"])
for it in range(3):
    dbg.nrf_dbg_bwd_trace(None, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if lm: L.check(lib.nrf_mlp_backward_f16_lm(m._m, P(feats), P(dirs), 192, P(gr), C.c_int64(p), P(g_blob), P(g_x), P(ws), C.c_size_t(nb), None))
    else: L.check(lib.nrf_mlp_backward_f16(m._m, P(x), P(gr), C.c_int64(p), P(g_blob), P(g_x), P(ws), C.c_size_t(nb), None))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
buf = (C.c_ulonglong * (256 * 16))(); dbg.nrf_dbg_bwd_trace(buf, 0)
a = np.array(buf[:], dtype=np.float64).reshape(256, 16)
tiles = p / 128 / 256                       # 128-point workgroup passes per workgroup
names = ["inputs (load, convert, store)", "forward (36 MFMA + 6 x store 4 fragments)", "g_out load", "colour last layer (transposes, dW, dX, mask)", "hidden colour x2: fragment reload", "  transposes (8 MFMA + conversions)", "  dW (8 MFMA)",
         "  dX gemm (8 MFMA)", "  mask -> fragments", "colour layer 0", "sigma last layer", "sigma hidden x1 (all of it)", "sigma layer 0 + g_x store", "(after the loop)", "whole kernel", "loop head"]
print(("level-major inputs; " if lm else "fp32 rows; ") + "call %.2f ms; %d points; per 128-point workgroup pass (wave 0), 100 MHz counter ticks x 24 = ~2.4 GHz cycles:" % (dt * 1e3, p))
for i, n in enumerate(names):
    if i == 14: continue
    print("  %-50s %8.1f ticks" % (n, a[:, i].sum() / 256 / tiles))
print("  whole kernel per workgroup: %.0f ticks = %.3f ms at 100 MHz" % (a[:, 14].sum() / 256, a[:, 14].sum() / 256 / 1e5))
