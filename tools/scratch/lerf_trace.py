"""diagnostic build (EXTRA=-DNRF_LERF_TRACE, NRF_LIB_PATH): where wave 0 of each workgroup of the geo-fed LeRF kernel B spends its cycles"""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, renderer as R, _lib as L
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
sc = S.make_lerf_scene(); r = sc["renderer"]
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
r.Render(800, 800, K, p, c2w=c2w, row0=0, rows=82); torch.cuda.synchronize()
lib = C.CDLL(os.environ["NRF_LIB_PATH"])
buf = (C.c_ulonglong * (256 * 8))()
lib.nrf_dbg_lerf_trace(None, 1)
t0 = time.perf_counter(); r.Render(800, 800, K, p, c2w=c2w); torch.cuda.synchronize(); dt = time.perf_counter() - t0
lib.nrf_dbg_lerf_trace(buf, 0)
a = np.array(buf[:], dtype=np.float64).reshape(256, 8)
it = a[:, 5].sum()
print("frame %.1f ms; kernel B iterations %d; per iteration (cycles):" % (dt * 1e3, int(it)))
for i, n in ((0, "tile loops"), (1, "hooks after a tile"), (2, "wait + barrier"), (7, "input loads"), (6, "norm + reduce tail"), (4, "iteration total")):
    print("  %-20s %9.0f" % (n, a[:, i].sum() / it))
print("  matrix-pipe cycles per iteration: LE0 8 x (12 + 8 x 2... ) = 8 x 28 x 32 + Gram 8 x 48 x 32 = %d" % (8 * 28 * 32 + 8 * 48 * 32))
