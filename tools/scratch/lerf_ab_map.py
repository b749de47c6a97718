"""LeRF frame time, alternating: sigma_le composed through the merge map (nrf_raw2weights_gather) vs gathered by torch first; same embedding"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, renderer as R, _lib as L
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
sc = S.make_lerf_scene(); r = sc["renderer"]
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
r.Render(800, 800, K, p, c2w=c2w, row0=0, rows=82); torch.cuda.synchronize()
ref = None
for rep in range(3):
    for mode in (True, False):
        r.compose_through_map = mode
        r.Render(800, 800, K, p, c2w=c2w); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): res = r.Render(800, 800, K, p, c2w=c2w)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
        e = res.Outputs.RenderedLangEmbedding
        if ref is None: ref = e.clone()
        print("through map" if mode else "torch gather", "ms/frame %.2f" % (dt * 1e3), "same embedding:", bool(torch.equal(e, ref)), flush=True)
