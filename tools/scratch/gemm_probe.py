"""nrf_gemm_nt_bf16x3 / nrf_gemm_nt_f16x3 against torch's fp32 matmul (rocBLAS): time and accuracy (against float64) at the training shapes, plus rows of wildly
different magnitudes (what a back-propagated gradient looks like) for the scaled fp16 arithmetic."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nerfpp_amd import _lib as L
torch.backends.cuda.matmul.allow_tf32 = False
lib = L.lib()


def call(fn, a, b, c, M, N, K):
    L.check(fn(C.c_void_p(a.data_ptr()), a.stride(0), C.c_int64(M), K, C.c_void_p(b.data_ptr()), b.stride(0), N, C.c_void_p(c.data_ptr()), N, None, 0, None))


for (M, N, K) in [(786432, 256, 256), (786432, 256, 319), (786432, 128, 283), (3145728 // 4, 256, 160), (786432, 33, 256), (1000, 70, 63)]:
    a = torch.randn((M, K), device="cuda"); b = torch.randn((N, K), device="cuda") * 0.1
    c1 = torch.empty((M, N), device="cuda"); c2 = torch.empty((M, N), device="cuda")
    runs = {"bf16x3": lambda: call(lib.nrf_gemm_nt_bf16x3, a, b, c1, M, N, K), "f16x3": lambda: call(lib.nrf_gemm_nt_f16x3, a, b, c2, M, N, K), "torch_fp32": lambda: a @ b.t()}
    out = {}
    for name, f in runs.items():
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        out[name + "_us"] = round((time.perf_counter() - t0) / 5 * 1e6, 1)
    want = (a.double() @ b.double().t())
    wm = want.abs().max()
    err = {"bf16x3": float((c1.double() - want).abs().max() / wm), "f16x3": float((c2.double() - want).abs().max() / wm), "fp32": float(((a @ b.t()).double() - want).abs().max() / wm)}
    print(json.dumps(dict(M=M, N=N, K=K, **out, max_err_over_max=err)), flush=True)

# rows of magnitudes 1e-12 .. 1e+8, zero rows, one huge entry in a row of small ones; weights x 1e-6 and x 1e+4: error PER ROW relative to the row's largest entry
M, N, K = 65536, 256, 256
g = torch.Generator(device="cuda"); g.manual_seed(7)
a = torch.randn((M, K), device="cuda", generator=g) * torch.pow(10.0, torch.rand((M, 1), device="cuda", generator=g) * 20 - 12)
a[::97] = 0.0
a[5::101, 3] *= 1e6
for wscale in (1.0, 1e-6, 1e4):
    b = torch.randn((N, K), device="cuda", generator=g) * 0.1 * wscale
    want = a.double() @ b.double().t()
    rowmax = want.abs().amax(dim=1).clamp_min(1e-300)
    res = {}
    for name, fn in (("bf16x3", lib.nrf_gemm_nt_bf16x3), ("f16x3", lib.nrf_gemm_nt_f16x3)):
        c = torch.full((M, N), float("nan"), device="cuda")
        call(fn, a, b, c, M, N, K)
        res[name] = float(((c.double() - want).abs().amax(dim=1) / rowmax).max())
        assert bool(torch.isfinite(c).all())
    res["fp32"] = float((((a @ b.t()).double() - want).abs().amax(dim=1) / rowmax).max())
    print(json.dumps(dict(case="rows 1e-12..1e+8, weights x %g" % wscale, worst_row_err_over_row_max=res)), flush=True)

# the weight-gradient product dW = G^T X (contraction over the points): nrf_gemm_tn_bf16x3 against torch's fp32 product
for (P, out, n) in [(786432, 256, 256), (786432, 256, 128), (786432, 128, 283), (786432, 768, 256)]:
    G = torch.randn((P, out), device="cuda") * 1e-3; X = torch.randn((P, n), device="cuda")
    dw = torch.zeros((out, n), device="cuda")
    runs = {"tn_bf16x3": lambda: L.check(lib.nrf_gemm_tn_bf16x3(C.c_void_p(G.data_ptr()), out, out, C.c_void_p(X.data_ptr()), n, n, C.c_int64(P), C.c_void_p(dw.data_ptr()), n, 0, None)),
            "torch_fp32": lambda: G.t() @ X}
    res = {}
    for name, f in runs.items():
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        res[name + "_us"] = round((time.perf_counter() - t0) / 5 * 1e6, 1)
    dw.zero_(); runs["tn_bf16x3"](); torch.cuda.synchronize()
    want = G.double().t() @ X.double()
    res["max_err_over_max"] = {"tn_bf16x3": float((dw.double() - want).abs().max() / want.abs().max()), "fp32": float(((G.t() @ X).double() - want).abs().max() / want.abs().max())}
    res["operand_GBps"] = round((P * (out + n) * 4) / (res["tn_bf16x3_us"] * 1e-6) / 1e9, 1)
    print(json.dumps(dict(product="dW = G^T X", P=P, out=out, n=n, **res)), flush=True)
