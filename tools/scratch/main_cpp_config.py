"""Frame time of the configuration the reference's main.cpp really runs (main.cpp:179-199, :227-231: CuHashEmbedder 16..1024, CuSHEncoder degree 8, NeRFSmall 3x64 + colour
3x64, 64 + 192 samples) beside BASELINE config 3 (16..512, degree 4, colour 4x64, 64 + 128): whole 800x800 frames, default precision, per-kernel ms per frame."""
import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S, renderer as R
H = W = 800
K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
lib = L.lib()
only = sys.argv[1] if len(sys.argv) > 1 else ""          # optional: substring selecting one configuration (profiling runs)
for name, kw, ni in (("BASELINE config 3 (16..512, SH 4, colour 4x64, 64+128)", dict(), 128),
                     ("main.cpp (16..1024, SH 8, colour 3x64, 64+192)", dict(finest=1024, sh_degree=8, num_layers_color=3), 192),
                     ("main.cpp grid only (16..1024, SH 4, colour 4x64, 64+128)", dict(finest=1024), 128)):
    if only and only not in name: continue
    sc = S.make_hash_scene(mode="cu", **kw)
    if os.environ.get("NRF_DENSE_GB"): sc["embedder"].set_dense_budget(int(float(os.environ["NRF_DENSE_GB"]) * (1 << 30)))          # e.g. 40: the 17 GB finest level of the 16..1024 grid baked too
    rp = S.lego_render_params(sc["bbox"], 64, ni, 65536, L.NRF_PREC_F16_SPLIT)
    r = sc["renderer"]
    for _ in range(3): r.Render(H, W, K, rp, c2w=c2w)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); out = r.Render(H, W, K, rp, c2w=c2w); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    L.check(lib.nrf_set_render_lanes(1))
    lib.nrf_profile_enable(1)
    ms = (C.c_double * len(L.NRF_PROF_NAMES))(); cnt = (C.c_int64 * len(L.NRF_PROF_NAMES))(); lib.nrf_profile_read(ms, cnt, 1)
    for _ in range(4): r.Render(H, W, K, rp, c2w=c2w)
    torch.cuda.synchronize()
    lib.nrf_profile_read(ms, cnt, 1); lib.nrf_profile_enable(0)
    L.check(lib.nrf_set_render_lanes(2))
    print("%-62s ms/frame min %.2f median %.2f (%.2e ray-samples/s) | one lane, per frame: " % (name, min(ts) * 1e3, sorted(ts)[3] * 1e3, H * W * (64 + 64 + ni) / min(ts))
          + ", ".join("%s %.2f" % (n, ms[i] / 4) for i, n in enumerate(L.NRF_PROF_NAMES) if ms[i] > 0), flush=True)
    del sc, r, out
    torch.cuda.empty_cache()
