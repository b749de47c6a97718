"""Does any render path read workspace it has not written?  Render, poison the renderer's cached workspace with 0xFF bytes (NaN as fp32 / fp16, -1 as an index), render
again: the outputs must not change.  usage (GPU box): python tools/scratch/poison_ws.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
lib = L.lib()
bad = 0
for name, sc in (("cu", S.make_hash_scene(mode="cu", log2_t=16)), ("ngp", S.make_hash_scene(mode="ngp", log2_t=16)), ("classic", S.make_classic_scene())):
    r = sc["renderer"]
    for prec in (L.NRF_PREC_F32, L.NRF_PREC_F16_MFMA, L.NRF_PREC_F16_SPLIT):
        for (h, w, chunk) in ((120, 97, 4000), (200, 237, 47400), (160, 100, 5001)):
            if name == "classic": h, w, chunk = h // 2, w // 2, chunk // 4
            for lanes in (1, 2):
                L.check(lib.nrf_set_render_lanes(lanes))
                K = S.lego_K(h, w); c2w = S.pose_spherical(40.0, -25.0, 4.0)
                rp = S.lego_render_params(sc["bbox"], 64, 128, chunk, prec, ReturnWeights=True)
                a = r.Render(h, w, K, rp, c2w=c2w).Outputs; a = [t.clone() for t in (a.RGBMap, a.DepthMap, a.AccMap, a.Weights)]
                torch.cuda.synchronize()
                r._ws.fill_(255); torch.cuda.synchronize()
                b = r.Render(h, w, K, rp, c2w=c2w).Outputs; b = [t.clone() for t in (b.RGBMap, b.DepthMap, b.AccMap, b.Weights)]
                torch.cuda.synchronize()
                same = all(torch.equal(x, y) for x, y in zip(a, b)); fin = all(bool(torch.isfinite(t).all()) for t in b)
                if not (same and fin):
                    bad += 1
                    d = [(float((x - y).abs().nan_to_num(1e9).max())) for x, y in zip(a, b)]
                    print(f"POISON SHOWS: {name} precision {prec} {h}x{w} chunk {chunk} lanes {lanes}: same {same} finite {fin} max diffs {d}", flush=True)
L.check(lib.nrf_set_render_lanes(2))
print("poisoned workspace changed", bad, "renders")
