"""One call each of the training GEMM kernels at 786 432 x 256 x 256 (for rocprofv3 --pmc passes: HBM bytes per launch)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L
lib = L.lib()
M, N, K = 786432, 256, 256
a = torch.randn((M, K), device="cuda"); b = torch.randn((N, K), device="cuda") * 0.1; c = torch.empty((M, N), device="cuda")
g = torch.randn((M, N), device="cuda") * 1e-3; dw = torch.zeros((N, K), device="cuda")
for _ in range(3):
    L.check(lib.nrf_gemm_nt_f16x3(C.c_void_p(a.data_ptr()), K, C.c_int64(M), K, C.c_void_p(b.data_ptr()), K, N, C.c_void_p(c.data_ptr()), N, None, 0, None))
    L.check(lib.nrf_gemm_nt_bf16x3(C.c_void_p(a.data_ptr()), K, C.c_int64(M), K, C.c_void_p(b.data_ptr()), K, N, C.c_void_p(c.data_ptr()), N, None, 0, None))
    L.check(lib.nrf_gemm_tn_bf16x3(C.c_void_p(g.data_ptr()), N, N, C.c_void_p(a.data_ptr()), K, K, C.c_int64(M), C.c_void_p(dw.data_ptr()), K, 0, None))
torch.cuda.synchronize()
print("operand bytes: NT A + C = %.3f GB, TN G + X = %.3f GB" % (2 * M * 256 * 4 / 1e9, 2 * M * 256 * 4 / 1e9))
