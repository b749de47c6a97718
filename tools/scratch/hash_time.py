"""frame time + per-kernel ms of the HashNeRF render for tuning builds (NRF_LIB_PATH); args: f16x3 | f16 [cu | ngp]"""
import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S, renderer as R
H = W = 800
prec = {"f16x3": L.NRF_PREC_F16_SPLIT, "f16": L.NRF_PREC_F16_MFMA}[sys.argv[1] if len(sys.argv) > 1 else "f16x3"]
sc = S.make_hash_scene(mode=sys.argv[2] if len(sys.argv) > 2 else "cu", table_amp=0.5, sigma_scale=30.0)
K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=int(os.environ.get("NRF_CHUNK", "131072")), Perturb=0.0, WhiteBkgr=True, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX, Precision=prec)
r = sc["renderer"]
for _ in range(3): r.Render(H, W, K, rp, c2w=c2w)
torch.cuda.synchronize()
lib = L.lib(); lib.nrf_profile_enable(1)
ms = (C.c_double * len(L.NRF_PROF_NAMES))(); cnt = (C.c_int64 * len(L.NRF_PROF_NAMES))(); lib.nrf_profile_read(ms, cnt, 1)
ts = []
for _ in range(8):
    t0 = time.perf_counter(); out = r.Render(H, W, K, rp, c2w=c2w); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
lib.nrf_profile_read(ms, cnt, 1); lib.nrf_profile_enable(0)
print("ms/frame min %.2f median %.2f | per frame: " % (min(ts) * 1e3, sorted(ts)[4] * 1e3) + ", ".join("%s %.2f" % (n, ms[i] / 8) for i, n in enumerate(L.NRF_PROF_NAMES)), "| rgb mean %.6f" % float(out.Outputs.RGBMap.mean()))
