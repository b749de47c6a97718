"""Do rays that miss the AABB get zero language density in the fused LeRF pass?  (debug probe)"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, renderer as R
sc = S.make_lerf_scene()
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
for fused in (True, False):
    sc["renderer"].fused = fused and sc["renderer"].fused
    p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=4096, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
    res = sc["renderer"].Render(800, 800, K, p, c2w=c2w, row0=0, rows=4)
    rays = res.Extras["rays_flat"].cpu().numpy(); acc = res.Outputs.AccMapLE.cpu().numpy()
    miss = rays[:, 7] <= rays[:, 6] + 2e-6
    print("fused", fused, "rays", len(acc), "miss", int(miss.sum()), "acc on missed rays: max", float(acc[miss].max()) if miss.any() else None, "acc overall min/max", float(acc.min()), float(acc.max()))
    if miss.any():
        i = np.nonzero(miss)[0][0]
        o, d, n, f = rays[i, :3], rays[i, 3:6], rays[i, 6], rays[i, 7]
        print(" first missed ray: near", n, "far", f, "point at near", o + d * n, "weights sum", float(res.Outputs.WeightsLE[i].sum()), "z_fine[0,-1]", res.Extras["z_fine"][i, 0].item(), res.Extras["z_fine"][i, -1].item())
