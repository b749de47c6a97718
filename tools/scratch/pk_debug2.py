import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, renderer as R, _lib as L
sc = S.make_hash_scene(mode="cu")
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
outs = {}
for lanes in (1, 2):
    L.check(L.lib().nrf_set_render_lanes(lanes))
    for chunk in (131072, 32768):
        rp = S.lego_render_params(sc["bbox"], chunk=chunk, precision=L.NRF_PREC_F16_SPLIT)
        for rep in range(2):
            res = sc["renderer"].Render(800, 800, K, rp, c2w=c2w)
            outs[(lanes, chunk, rep)] = res.Outputs.RGBMap.reshape(-1, 3).cpu()
torch.save(outs, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])[(1, 32768, 0)]
    for k, v in outs.items():
        d = (v != ref).any(1)
        rows = torch.nonzero(d).flatten() // 800
        print(k, "differing pixels", int(d.sum()), "rows", (int(rows.min()), int(rows.max())) if d.any() else None,
              "by 100-row band", torch.bincount(rows // 100, minlength=8).tolist() if d.any() else None, "max abs", float((v - ref).abs().max()))
else:
    ref = outs[(1, 32768, 0)]
    for k, v in outs.items(): print(k, "self-consistent", bool(torch.equal(v, ref)))
