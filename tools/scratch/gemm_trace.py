"""diagnostic build (-DNRF_GB_TRACE, NRF_LIB_PATH): where waves 0 and 7 of each workgroup of the whole-row GEMM kernel (k_gemm_nt_rows, NRF_GEMM_CREWS=0) spend
their time, per section and output tile (shader-clock stamps, scaled to us by the call's measured duration)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L
lib = L.lib(); dbg = C.CDLL(os.environ["NRF_LIB_PATH"])
M, N, K = 786432, 256, 256
a = torch.randn((M, K), device="cuda"); b = torch.randn((N, K), device="cuda") * 0.1; c = torch.empty((M, N), device="cuda")
names = {0: "tile head (pointers)", 1: "row maxima = wait for the whole A block (F16)", 2: "B tiles 0, 1 requested; A / B tile 0 -> LDS", 3: "barrier waits (9 per tile)", 4: "multiply x8 (fragment reads + 24 MFMA each)",
         5: "A / B tile k + 1 -> LDS, next tile's A requested (x8)", 6: "accumulators -> LDS", 7: "barrier", 8: "C rows LDS -> memory", 9: "barrier"}
for name, fn in (("f16x3", lib.nrf_gemm_nt_f16x3), ("bf16x3", lib.nrf_gemm_nt_bf16x3)):
    call = lambda: L.check(fn(C.c_void_p(a.data_ptr()), K, C.c_int64(M), K, C.c_void_p(b.data_ptr()), K, N, C.c_void_p(c.data_ptr()), N, None, 0, None))
    call(); torch.cuda.synchronize()
    dbg.nrf_dbg_gb_trace(None, 1)
    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    buf = (C.c_ulonglong * (256 * 2 * 16))(); dbg.nrf_dbg_gb_trace(buf, 0)
    tr = np.array(buf[:], dtype=np.float64).reshape(256, 2, 16)
    tiles = M / 128 / 256
    print("%s: call %.0f us (with the stamps) = %.2f us per output tile and workgroup:" % (name, dt * 1e6, dt * 1e6 / tiles))
    for w, wn in ((0, "wave 0"), (1, "wave 7")):
        whole = tr[:, w, 15].mean()
        print("  %s (%.0f ticks per workgroup):" % (wn, whole))
        for i in range(10):
            f = tr[:, w, i].mean() / whole
            print("    %-62s %5.1f %%  %6.2f us" % (names[i], 100 * f, f * dt * 1e6 / tiles))
