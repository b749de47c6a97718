"""How much of the fine pass's work lands on samples whose compositing weight is negligible (bench scene, one 100-row band)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, _lib as L
sc = S.make_hash_scene(mode="cu")
K = S.lego_K(800, 800); c2w = S.pose_spherical(-180.0, -30.0, 4.0)
rp = S.lego_render_params(sc["bbox"], 64, 128, 131072, L.NRF_PREC_F16_SPLIT, ReturnWeights=True)
res = sc["renderer"].Render(800, 800, K, rp, c2w=c2w, row0=350, rows=100)
w = res.Outputs.Weights
n, s = w.shape
for thr in (0.0, 1e-12, 1e-9, 1e-7, 1e-5):
    keep = w > thr
    last = torch.where(keep.any(1), (keep.float() * torch.arange(1, s + 1, device=w.device)).amax(1), torch.zeros(n, device=w.device))
    tiles32 = keep.reshape(n, s // 32, 32).any(2).float().mean()
    tiles64 = keep.reshape(n, s // 64, 64).any(2).float().mean()
    print("thr %.0e: samples kept %.3f, mean last kept index %.1f of %d, 32-tiles with any kept %.3f, 64-tiles %.3f, rays with none %.3f" %
          (thr, float(keep.float().mean()), float(last.mean()), s, float(tiles32), float(tiles64), float((~keep.any(1)).float().mean())))
print("acc mean %.3f" % float(res.Outputs.AccMap.mean()))
