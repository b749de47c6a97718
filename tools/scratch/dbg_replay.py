import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L
lib = L.lib()
A = torch.from_numpy(np.fromfile("gpurun_out/dbg_A.bin", np.float32).reshape(4096, 384)).cuda()
B = torch.from_numpy(np.fromfile("gpurun_out/dbg_B.bin", np.float32).reshape(256, 256)).cuda()
D = np.fromfile("gpurun_out/dbg_D.bin", np.float32).reshape(4096, 384)[:, :256]
Hm = np.fromfile("gpurun_out/dbg_H.bin", np.float32).reshape(4096, 384)[:, :256]
want = (A[:, :256].double() @ B.double().t()).cpu().numpy() * (Hm > 0)
print("dumped dst: nonfinite", int((~np.isfinite(D)).sum()), "max |dst - want| / max|want| over finite", float(np.nanmax(np.abs(np.where(np.isfinite(D), D, 0) - want)) / np.abs(want).max()))
bad = np.argwhere(~np.isfinite(D))
print("first bad", bad[:5].tolist(), "bad columns", sorted(set(bad[:, 1].tolist()))[:40], "bad rows count", len(set(bad[:, 0].tolist())))
c = torch.empty((4096, 256), device="cuda")
for name, fn in (("f16x3", lib.nrf_gemm_nt_f16x3), ("bf16x3", lib.nrf_gemm_nt_bf16x3)):
    L.check(fn(C.c_void_p(A.data_ptr()), 384, C.c_int64(4096), 256, C.c_void_p(B.data_ptr()), 256, 256, C.c_void_p(c.data_ptr()), 256, None, 0, None))
    got = c.cpu().numpy() * (Hm > 0)
    print(name, "replay: nonfinite", int((~np.isfinite(got)).sum()), "err", float(np.abs(got - want).max() / np.abs(want).max()))
print("A stats: finite", bool(torch.isfinite(A[:, :256]).all()), "absmax", float(A[:, :256].abs().max()), "min nonzero", float(A[:, :256][A[:, :256] != 0].abs().min()))
print("B stats: absmax", float(B.abs().max()))
