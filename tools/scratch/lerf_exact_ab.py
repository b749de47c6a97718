"""LeRF split-precision frame with / without the exact-fp32 coarse sigma pass (sigma_lerf_f32.hip) and with / without its geo hand-over."""
import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, renderer as R, _lib as L
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
sc = S.make_lerf_scene(); r = sc["renderer"]
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
r.set_precision(L.NRF_PREC_F16_SPLIT)
lib = L.lib()
for exact, hand in ((True, True), (True, False), (False, True)):
    r.exact_coarse, r.hand_over_geo = exact, hand
    r.Render(800, 800, K, p, c2w=c2w, row0=0, rows=41); torch.cuda.synchronize()
    lib.nrf_profile_enable(1); ms = (C.c_double * len(L.NRF_PROF_NAMES))(); cnt = (C.c_int64 * len(L.NRF_PROF_NAMES))(); lib.nrf_profile_read(ms, cnt, 1)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); res = r.Render(800, 800, K, p, c2w=c2w); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    lib.nrf_profile_read(ms, cnt, 1); lib.nrf_profile_enable(0)
    print("exact_coarse", exact, "hand_over_geo", hand, "s/frame %.4f" % min(ts), {n: round(ms[i] / 3, 1) for i, n in enumerate(L.NRF_PROF_NAMES)}, {n: cnt[i] // 3 for i, n in enumerate(L.NRF_PROF_NAMES)}, flush=True)
