"""Frames far larger than the bench's: 3000 x 2400 (7.2 M rays, 1.8 G ray-samples) -- index arithmetic past 2^31 sample slots per frame, hundreds of chunks; the last rows must
equal the same rows rendered as a tile, everything finite.  Also a LeRF frame at 1600 x 1200.  usage (GPU box): python tools/scratch/big_frame.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
bad = 0
for mode in ("cu", "ngp"):
    sc = S.make_hash_scene(mode=mode); r = sc["renderer"]
    h, w = 2400, 3000
    K = S.lego_K(h, w); c2w = S.pose_spherical(30.0, -30.0, 4.0)
    for prec, chunk in ((L.NRF_PREC_F16_SPLIT, 65536), (L.NRF_PREC_F32, 32768)):
        if prec == L.NRF_PREC_F32: h2, w2 = 1200, 1000
        else: h2, w2 = h, w
        K2 = S.lego_K(h2, w2)
        rp = S.lego_render_params(sc["bbox"], 64, 128, chunk, prec, ReturnWeights=False)
        t0 = time.perf_counter(); full = r.Render(h2, w2, K2, rp, c2w=c2w).Outputs; torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ok = bool(torch.isfinite(full.RGBMap).all() and torch.isfinite(full.DepthMap).all())
        row0 = h2 - 37
        t = r.Render(h2, w2, K2, rp, c2w=c2w, row0=row0, rows=37).Outputs
        same = torch.equal(t.RGBMap, full.RGBMap[row0:]) and torch.equal(t.DepthMap, full.DepthMap[row0:])
        t2 = r.Render(h2, w2, K2, rp, c2w=c2w, row0=h2 // 2, rows=3).Outputs
        same = same and torch.equal(t2.RGBMap, full.RGBMap[h2 // 2:h2 // 2 + 3])
        bad += not (ok and same)
        print(f"{mode} precision {prec} {h2}x{w2} = {h2 * w2} rays: {dt * 1e3:.0f} ms, finite {ok}, tiles == rows of the frame {same}, mean acc {float(full.AccMap.mean()):.3f}", flush=True)
    del sc, r
    torch.cuda.empty_cache()
lsc = S.make_lerf_scene(); lr = lsc["renderer"]
h, w = 1200, 1600
p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=32768, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=lsc["bbox"])      # (without ReturnWeights the reference drops the rendered embedding too, LeRFRenderer.cpp:180-185)
K = S.lego_K(h, w); c2w = S.pose_spherical(30.0, -30.0, 4.0)
t0 = time.perf_counter(); full = lr.Render(h, w, K, p, c2w=c2w).Outputs.RenderedLangEmbedding; torch.cuda.synchronize(); dt = time.perf_counter() - t0
t = lr.Render(h, w, K, p, c2w=c2w, row0=h - 5, rows=5).Outputs.RenderedLangEmbedding
ok = bool(torch.isfinite(full).all()); same = torch.equal(t.reshape(5, w, -1), full.reshape(h, w, -1)[h - 5:])
bad += not (ok and same)
print(f"lerf {h}x{w}: {dt * 1e3:.0f} ms, finite {ok}, tile == rows {same}", flush=True)
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
