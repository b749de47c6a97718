import numpy as np, torch, sys
sys.path.insert(0, ".")
from nerfpp_amd import _lib as L, scene as S
sc = S.make_hash_scene(mode="cu")
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
out = {}
for prec in (L.NRF_PREC_F32, L.NRF_PREC_F16_SPLIT, L.NRF_PREC_F16_MFMA):
    rp = S.lego_render_params(sc["bbox"], chunk=4096, precision=prec, KeepIntermediates=True, ReturnRaw=True)
    out[prec] = sc["renderer"].Render(800, 800, K, rp, c2w=c2w, row0=380, rows=40)
a, b, c = out[0], out[2], out[1]
h = lambda t: t.cpu().numpy()
for name, x in (("split", b), ("f16", c)):
    d = np.abs(h(x.Outputs.RGBMap) - h(a.Outputs.RGBMap)).reshape(-1)
    print(name, "quantiles 50/90/99/99.9/max", np.quantile(d, [0.5, 0.9, 0.99, 0.999]), d.max(), "frac<1e-4", (d < 1e-4).mean(), "psnr", S.psnr(h(x.Outputs.RGBMap), h(a.Outputs.RGBMap)))
    dr = np.abs(h(x.Extras["raw_coarse"]) - h(a.Extras["raw_coarse"]))
    print("  raw coarse max", dr.max(), "scale", np.abs(h(a.Extras["raw_coarse"])).max(), "wc max diff", np.abs(h(x.Extras["weights_coarse"]) - h(a.Extras["weights_coarse"])).max())
    dz = np.abs(h(x.Extras["z_fine"]) - h(a.Extras["z_fine"])).max(axis=1)
    print("  rays with all fine z within 1e-6:", (dz < 1e-6).mean(), "1e-4:", (dz < 1e-4).mean())
    m = dz < 1e-6
    dd = np.abs(h(x.Outputs.RGBMap).reshape(-1, 3) - h(a.Outputs.RGBMap).reshape(-1, 3))[m]
    print("  pixel err on those rays: max", dd.max(), "q99", np.quantile(dd, 0.99))
