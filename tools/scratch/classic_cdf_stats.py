"""Classic NeRF: how far the split-precision coarse CDF is from the fp32 one, per ray, and how close the fine pass's u values come to a CDF edge --
the two quantities certified sampling weighs against each other."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, _lib as L
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sc = S.make_classic_scene()
K = S.lego_K(800, 800); c2w = S.pose_spherical(-180.0, -30.0, 4.0)
out = {}
for name, prec in (("f32", L.NRF_PREC_F32), ("split", L.NRF_PREC_F16_SPLIT)):
    rp = S.lego_render_params(sc["bbox"], 64, 128, 8192, prec, KeepIntermediates=True)
    res = sc["renderer"].Render(800, 800, K, rp, c2w=c2w, row0=(800 - rows) // 2, rows=rows)
    out[name] = {k: v.double() for k, v in res.Extras.items() if k in ("weights_coarse", "z_fine", "raw_coarse", "z_coarse")}
    out[name]["rgb"] = res.Outputs.RGBMap.double()
def cdf(w):
    w = w[:, 1:-1] + 1e-8
    pdf = w / w.sum(-1, keepdim=True)
    return torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
ca, cb = cdf(out["f32"]["weights_coarse"]), cdf(out["split"]["weights_coarse"])
dc = (ca - cb).abs().max(1).values
sa, sb = out["f32"]["raw_coarse"][..., 3], out["split"]["raw_coarse"][..., 3]
print("rays", ca.shape[0], "sigma scale", float(sa.abs().max()), "max |dsigma|", float((sa - sb).abs().max()), "median |dsigma|", float((sa - sb).abs().median()))
rel = ((sa - sb).abs() / (sa.abs() + 1e-3)).flatten()
print("relative dsigma: median %.3e  99.9%% %.3e max %.3e" % (float(rel.median()), float(rel.kthvalue(int(0.999 * rel.numel())).values), float(rel.max())))
print("max |dcdf| per ray: median %.3e  99%% %.3e  99.9%% %.3e  max %.3e" % (float(dc.median()), float(dc.kthvalue(int(0.99 * dc.numel())).values), float(dc.kthvalue(int(0.999 * dc.numel())).values), float(dc.max())))
u = torch.linspace(0.0, 1.0, 128, dtype=torch.float32).double().cuda()
marg = (ca[:, 1:-1, None] - u[None, None, :]).abs().amin((1, 2))          # interior edges only
zf_a, zf_b = out["f32"]["z_fine"], out["split"]["z_fine"]
mism = ((zf_a - zf_b).abs().max(1).values > 1e-5)
exact_rows = (zf_a == zf_b).all(1)
print("rays with a different sample set (|dz| > 1e-5): %d (%.4f %%), rays with bit-identical z_fine: %.2f %%" % (int(mism.sum()), 100 * float(mism.double().mean()), 100 * float(exact_rows.double().mean())))
print("margin of the mismatching rays: max %.3e ; their |dcdf|: min %.3e" % (float(marg[mism].max()) if mism.any() else -1, float(dc[mism].min()) if mism.any() else -1))
for eps in (1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4):
    fl = marg <= eps
    print("eps %.0e: flagged %.3f %%, mismatching rays not flagged: %d" % (eps, 100 * float(fl.double().mean()), int((mism & ~fl).sum())))
d = (out["f32"]["rgb"] - out["split"]["rgb"]).abs()
print("pixels: max %.3e, frac within 1e-4 %.5f" % (float(d.max()), float((d < 1e-4).double().mean())))
