import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nerfpp_amd import _lib as L, scene as S, renderer as R, modules as M, synth
Lv, F, T = 16, 8, 19
bbox = S.LEGO_BBOX
e = M.CuHashEmbedder("lang_embedder", bbox, Lv, F, T, 16, 1024)
e.set_primes(np.array(S.CU_PRIMES[:3 * Lv], np.int32))
e.set_table(synth.synth_sym(311, (Lv * (1 << T) * F,), np.float32(0.5)))
shapes = [("s0", (256, 128)), ("s1", (33, 256)), ("l0", (256, 160)), ("l1", (768, 256))]
blob = np.concatenate([synth.synth_sym(900 + i, (np.prod(s),), np.float32(1.6 * np.sqrt(6.0 / sum(s)))) for i, (n, s) in enumerate(shapes)])
lerf = M.LeRF(32, 2, 256, 768, 128, "lang_model", params=blob)
r = R.LeRFRenderer(e, lerf)
K = S.lego_K(800, 800); c2w = S.pose_spherical(40.0, -30.0, 4.0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=False, ThinRay=True, BoundingBox=bbox)
r.Render(800, 800, K, p, c2w=c2w, row0=400, rows=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = r.Render(800, 800, K, p, c2w=c2w, row0=400, rows=rows)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = rows * 800
print("LeRF render: %d rays, %.3f s, %.3e ray-samples/s, full frame %.1f s" % (n, dt, n * 256 / dt, dt * 800 / rows))
