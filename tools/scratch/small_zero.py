"""NeRFSmall split kernel on the bench frame with its real (synthetic) weights vs all-zero weights: same instruction stream, same cycle count; what changes is the clock
the chip holds (DVFS give-back, MI355X_MICROARCH.md).  Prints per-stage ms per frame."""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S, renderer as R, modules as M
H = W = 800
K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
for tag in ("synthetic weights", "all-zero weights"):
    sc = S.make_hash_scene(mode="cu")
    r = sc["renderer"]
    if tag.startswith("all-zero"):
        mlp = M.NeRFSmall(3, 64, 15, 4, 64, False, 3, 64, 32, 16, "model", params=np.zeros_like(sc["mlp_blob"]))
        r = R.NeRFRenderer(sc["embedder"], sc["embeddirs"], mlp)
    rp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=131072, Perturb=0.0, WhiteBkgr=True, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT)
    for _ in range(3): r.Render(H, W, K, rp, c2w=c2w)
    torch.cuda.synchronize()
    lib = L.lib(); lib.nrf_profile_enable(1)
    ms = (C.c_double * len(L.NRF_PROF_NAMES))(); cnt = (C.c_int64 * len(L.NRF_PROF_NAMES))(); lib.nrf_profile_read(ms, cnt, 1)
    for _ in range(8): r.Render(H, W, K, rp, c2w=c2w)
    torch.cuda.synchronize()
    lib.nrf_profile_read(ms, cnt, 1); lib.nrf_profile_enable(0)
    print("%-18s" % tag, ", ".join("%s %.2f" % (n, ms[i] / 8) for i, n in enumerate(L.NRF_PROF_NAMES)), flush=True)
