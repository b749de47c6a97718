import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import scene as S, renderer as R, _lib as L
sc = S.make_hash_scene(mode="cu")
K = S.lego_K(800, 800); c2w = S.pose_spherical(30.0, -30.0, 4.0)
L.check(L.lib().nrf_set_render_lanes(1))
rp = S.lego_render_params(sc["bbox"], chunk=32768, precision=L.NRF_PREC_F16_SPLIT, ReturnRaw=True, KeepIntermediates="depths")
res = sc["renderer"].Render(800, 800, K, rp, c2w=c2w, row0=300, rows=41)
out = {"raw": res.Raw.cpu(), "z": res.Extras["z_fine"].cpu(), "zc": res.Extras["z_coarse"].cpu(), "rays": res.Extras["rays_flat"].cpu()}
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    a = torch.load(sys.argv[2]); b = out
    print("z_fine equal:", bool(torch.equal(a["z"], b["z"])))
    d = (a["raw"] != b["raw"]).any(-1)
    print("differing samples:", int(d.sum()), "of", d.numel(), " rays affected:", int(d.any(1).sum()))
    rays_bad = torch.nonzero(d.any(1)).flatten()
    print("first affected rays:", rays_bad[:20].tolist())
    for r in rays_bad[:6].tolist():
        js = torch.nonzero(d[r]).flatten().tolist()
        coarse = torch.isin(b["z"][r], b["zc"][r])
        print(" ray", r, "bad sample slots", js[:24], "... coarse flags", [int(coarse[j]) for j in js[:24]])
    # histogram of bad slots modulo 64 and of ray index modulo chunk positions
    sl = torch.nonzero(d)[:, 1]
    print("bad slot histogram (by 16):", torch.bincount(sl // 16, minlength=12).tolist())
    print("bad ray index mod 512 histogram (by 64):", torch.bincount((torch.nonzero(d)[:, 0] % 512) // 64, minlength=8).tolist())
