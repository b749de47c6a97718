"""How many of the fine pass's 192 samples per ray carry a compositing weight of EXACTLY zero (sigma <= 0, or transmittance underflowed to 0) on the bench scene?
Their colour cannot reach any output (0 * finite): the colour net's evaluation there is dead work."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S
H = W = 800
for name, sc in (("hash (bench scene)", S.make_hash_scene(mode="cu")), ("classic (bench scene)", S.make_classic_scene())):
    K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
    rp = S.lego_render_params(sc["bbox"], 64, 128, 16384, L.NRF_PREC_F16_SPLIT, ReturnWeights=True)
    res = sc["renderer"].Render(H, W, K, rp, c2w=c2w, row0=300, rows=200)
    w = res.Outputs.Weights
    z = (w == 0).float().mean().item(); tiny = (w.abs() < 1e-12).float().mean().item(); t7 = (w.abs() < 1e-7).float().mean().item()
    per_ray = (w != 0).sum(-1).float()
    print("%-22s weights %s: exactly zero %.3f, |w| < 1e-12 %.3f, |w| < 1e-7 %.3f; non-zero samples per ray: mean %.1f, max %d; acc mean %.3f" %
          (name, tuple(w.shape), z, tiny, t7, per_ray.mean().item(), int(per_ray.max().item()), res.Outputs.AccMap.mean().item()), flush=True)
