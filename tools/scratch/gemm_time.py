"""Time of nrf_gemm_nt_f16x3 / nrf_gemm_nt_bf16x3 at one shape (M N K), 20 calls each (tools/scratch/gemm_ablate.sh: timing-only builds under NRF_LIB_PATH)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L
lib = L.lib()
M, N, K = (int(x) for x in sys.argv[1:4])
a = torch.randn((M, K), device="cuda"); b = torch.randn((N, K), device="cuda") * 0.1; c = torch.empty((M, N), device="cuda")
out = []
for name, fn in (("f16x3", lib.nrf_gemm_nt_f16x3), ("bf16x3", lib.nrf_gemm_nt_bf16x3)):
    call = lambda: L.check(fn(C.c_void_p(a.data_ptr()), K, C.c_int64(M), K, C.c_void_p(b.data_ptr()), K, N, C.c_void_p(c.data_ptr()), N, None, 0, None))
    for _ in range(3):
        call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    out.append("%s %.1f us" % (name, (time.perf_counter() - t0) / 20 * 1e6))
print("  ".join(out))
