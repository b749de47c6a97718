"""Does a CU-masked stream (hipExtStreamCreateWithCUMask) work here, and how do the LeRF frame's two kinds of kernels scale with the CUs they get?
The F = 8 hash encode is HBM-bound (0.87 of the peak): if a part of the chip saturates the memory system, the matrix kernels could have the rest at the same time.
usage (GPU box): python tools/scratch/cu_mask_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S
hip = C.CDLL("libamdhip64.so")
lib = L.lib()
sc = S.make_lerf_scene()
h, m = sc["embedder"]._h, sc["lerf"]._m
P = lambda t: C.c_void_p(t.data_ptr())

def masked(pattern):
    """pattern(i) -> bool for CU bit i (256 bits)"""
    words = [0] * 8
    for i in range(256):
        if pattern(i): words[i // 32] |= 1 << (i % 32)
    arr = (C.c_uint32 * 8)(*words); s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return s

n_rays, s = 32768, 64
nc = n_rays * s
rng = np.random.default_rng(3)
bb = np.asarray(sc["bbox"], np.float32)
pts = torch.from_numpy(rng.uniform(bb[:3], bb[3:], (nc, 3)).astype(np.float32)).cuda()
x = torch.empty((16, nc, 8), dtype=torch.float16, device="cuda"); keep = torch.empty((nc,), dtype=torch.uint8, device="cuda")
sig = torch.empty((nc,), device="cuda"); geo = torch.empty((int(lib.nrf_lerf_geo_bytes(C.c_int64(nc))),), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()

def timed(f, st, reps=5):
    f(st); hip.hipStreamSynchronize(st)
    t0 = time.perf_counter()
    for _ in range(reps): f(st)
    hip.hipStreamSynchronize(st)
    return (time.perf_counter() - t0) / reps * 1e3

enc = lambda st: L.check(lib.nrf_hash_encode_lm_f16_strided(h, P(pts), C.c_int64(nc), P(x), C.c_int64(nc), P(keep), st))
sgm = lambda st: L.check(lib.nrf_lerf_sigma_exact_lm_strided(m, P(x), C.c_int64(nc), P(keep), C.c_int64(nc), P(sig), P(geo), C.c_int64(nc), st))
pats = [("all 256", lambda i: True), ("128: even bits", lambda i: i % 2 == 0), ("128: low half", lambda i: i < 128), ("128: high half", lambda i: i >= 128),
        ("64: low", lambda i: i < 64), ("96: low", lambda i: i < 96), ("160: high", lambda i: i >= 96), ("192: high", lambda i: i >= 64), ("32: low", lambda i: i < 32),
        ("bits 0-15 of every 32", lambda i: i % 32 < 16), ("bits 0-7 of every 32", lambda i: i % 32 < 8)]
streams = {}
for name, pat in pats:
    st = masked(pat); streams[name] = st
    print(f"{name:<16} F=8 encode of {nc} points {timed(enc, st):7.3f} ms    exact sigma_le kernel {timed(sgm, st):7.3f} ms", flush=True)
# both kinds at once on complementary masks against back to back on the whole chip
def pair(na, pa, nb, pb):
    a, b = masked(pa), masked(pb)
    def both():
        enc(a); sgm(b)
        hip.hipStreamSynchronize(a); hip.hipStreamSynchronize(b)
    both(); t0 = time.perf_counter()
    for _ in range(5): both()
    print(f"encode on {na} || sigma on {nb}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per pair", flush=True)
pair("low 128", lambda i: i < 128, "high 128", lambda i: i >= 128)
pair("low 96", lambda i: i < 96, "high 160", lambda i: i >= 96)
pair("low 64", lambda i: i < 64, "high 192", lambda i: i >= 64)
pair("bits 0-11 of every 32", lambda i: i % 32 < 12, "bits 12-31 of every 32", lambda i: i % 32 >= 12)
pair("all", lambda i: True, "all (two unmasked streams)", lambda i: True)
allst = streams["all 256"]
def serial():
    enc(allst); sgm(allst); hip.hipStreamSynchronize(allst)
serial(); t0 = time.perf_counter()
for _ in range(5): serial()
print(f"back to back on the whole chip: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per pair", flush=True)
