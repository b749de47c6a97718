"""frame time of the classic NeRF render (no checks): for tuning builds selected with NRF_LIB_PATH"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S, renderer as R
H = W = 800
sc = S.make_classic_scene()
K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=131072, Perturb=0.0, WhiteBkgr=True, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX,
                        Precision=L.NRF_PREC_F16_MFMA)
r = sc["renderer"]
for _ in range(2): r.Render(H, W, K, rp, c2w=c2w)
torch.cuda.synchronize(); ts = []
for _ in range(5):
    t0 = time.perf_counter(); r.Render(H, W, K, rp, c2w=c2w); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("ms/frame min %.2f median %.2f" % (min(ts) * 1e3, sorted(ts)[2] * 1e3))
