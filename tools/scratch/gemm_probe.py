"""nrf_gemm_nt_bf16x3 against torch's fp32 matmul (rocBLAS): time and accuracy at the training shapes."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L
torch.backends.cuda.matmul.allow_tf32 = False
lib = L.lib()
for (M, N, K) in [(786432, 256, 256), (786432, 256, 319), (786432, 128, 283), (3145728 // 4, 256, 160), (786432, 33, 256), (1000, 70, 63)]:
    a = torch.randn((M, K), device="cuda"); b = torch.randn((N, K), device="cuda") * 0.1; c = torch.empty((M, N), device="cuda")
    def mine():
        L.check(lib.nrf_gemm_nt_bf16x3(C.c_void_p(a.data_ptr()), K, C.c_int64(M), K, C.c_void_p(b.data_ptr()), K, N, C.c_void_p(c.data_ptr()), N, None, 0, None))
    def ref():
        return a @ b.t()
    for f in (mine, ref):
        f(); torch.cuda.synchronize()
    out = {}
    for name, f in (("bf16x3", mine), ("torch_fp32", ref)):
        t0 = time.perf_counter()
        for _ in range(5):
            r = f()
        torch.cuda.synchronize()
        out[name + "_us"] = (time.perf_counter() - t0) / 5 * 1e6
    want = (a.double() @ b.double().t())
    e1 = float((c.double() - want).abs().max() / want.abs().max()); e2 = float(((a @ b.t()).double() - want).abs().max() / want.abs().max())
    fl = 2.0 * M * N * K
    print(json.dumps(dict(M=M, N=N, K=K, **{k: round(v, 1) for k, v in out.items()}, bf16x3_tflops_fp32_equiv=round(fl / out["bf16x3_us"] / 1e6, 1),
                          torch_tflops=round(fl / out["torch_fp32_us"] / 1e6, 1), max_err_over_max_bf16x3=e1, max_err_over_max_fp32=e2)), flush=True)
