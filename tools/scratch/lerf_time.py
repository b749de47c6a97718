"""LeRF render pass timing (bench.py's lerf_measurement) with the level-major fp16 features and with fp32 rows"""
import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from nerfpp_amd import scene as S, renderer as R, _lib as L
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
for lm in (True, False, True):
    sc = S.make_lerf_scene(); r = sc["renderer"]
    if not lm: r.level_major = False
    p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
    r.Render(800, 800, K, p, c2w=c2w, row0=0, rows=41); torch.cuda.synchronize()
    lib = L.lib(); lib.nrf_profile_enable(1); ms = (C.c_double * len(L.NRF_PROF_NAMES))(); cnt = (C.c_int64 * len(L.NRF_PROF_NAMES))(); lib.nrf_profile_read(ms, cnt, 1)
    t0 = time.perf_counter(); res = r.Render(800, 800, K, p, c2w=c2w, row0=300, rows=200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lib.nrf_profile_read(ms, cnt, 1); lib.nrf_profile_enable(0)
    print("level_major", r.level_major, "s/frame %.3f" % (dt * 4), "units/s %.3e" % (160000 * 256 / dt), {n: round(ms[i] * 4, 1) for i, n in enumerate(L.NRF_PROF_NAMES)}, float(res.Outputs.RenderedLangEmbedding.abs().mean()))
