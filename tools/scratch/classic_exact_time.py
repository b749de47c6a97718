"""Classic NeRF frame: coarse pass = density branch in exact fp32 on the matrix cores (NRF_COARSE_AUTO) vs the whole network in split arithmetic (NRF_COARSE_FULL)."""
import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, _lib as L
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
sc = S.make_classic_scene()
lib = L.lib()
for name, cm in (("auto (exact sigma coarse)", L.NRF_COARSE_AUTO), ("full (split coarse)", L.NRF_COARSE_FULL)):
    rp = S.lego_render_params(sc["bbox"], 64, 128, 8192, L.NRF_PREC_F16_SPLIT, CoarseMode=cm)
    sc["renderer"].Render(800, 800, K, rp, c2w=c2w, row0=0, rows=41); torch.cuda.synchronize()
    lib.nrf_profile_enable(1); ms = (C.c_double * len(L.NRF_PROF_NAMES))(); cnt = (C.c_int64 * len(L.NRF_PROF_NAMES))(); lib.nrf_profile_read(ms, cnt, 1)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); sc["renderer"].Render(800, 800, K, rp, c2w=c2w); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    lib.nrf_profile_read(ms, cnt, 1); lib.nrf_profile_enable(0)
    print(name, "s/frame %.4f" % min(ts), {n: round(ms[i] / 3, 1) for i, n in enumerate(L.NRF_PROF_NAMES)}, flush=True)
