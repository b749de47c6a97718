"""Several renderer objects of one process rendering at the same time on different streams (each has its own lanes and workspace): every frame equals the one rendered alone,
bit for bit; also a trainer stepping on one stream while another scene renders on a second.  usage (GPU box): python tools/scratch/concurrency_fuzz.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
from nerfpp_amd.train import Trainer
rng = np.random.default_rng(5)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
scenes = [S.make_hash_scene(mode="cu", log2_t=15, seed=1), S.make_hash_scene(mode="ngp", log2_t=15, seed=2), S.make_classic_scene(), S.make_hash_scene(mode="cu", log2_t=15, seed=3)]
lerf = S.make_lerf_scene(log2_t=14)
tsc = S.make_hash_scene(mode="cu", log2_t=14, table_amp=1e-2, sigma_scale=4.0, seed=9)
tr = Trainer(tsc["embedder"], tsc["embeddirs"], tsc["mlp"], tsc["table"], tsc["mlp_blob"], mlp_backward="f16", hash_backward="binned")
streams = [torch.cuda.Stream() for _ in range(6)]
bad = 0
for rd in range(rounds):
    jobs = []
    for i, sc in enumerate(scenes):
        h, w = (int(rng.integers(60, 260)), int(rng.integers(60, 260))) if i != 2 else (int(rng.integers(20, 60)), int(rng.integers(20, 60)))
        chunk = int(rng.choice([4096, 32768, 65536, h * w]))
        prec = int(rng.choice([L.NRF_PREC_F16_SPLIT, L.NRF_PREC_F16_MFMA, L.NRF_PREC_F32]))
        K = S.lego_K(h, w); c2w = S.pose_spherical(float(rng.uniform(-180, 180)), -30.0, 4.0)
        jobs.append((sc["renderer"], h, w, K, S.lego_render_params(sc["bbox"], 64, 128, chunk, prec), c2w))
    alone = []
    for (r, h, w, K, rp, c2w) in jobs:
        alone.append(r.Render(h, w, K, rp, c2w=c2w).Outputs.RGBMap.clone())
    lh, lw = int(rng.integers(30, 90)), int(rng.integers(30, 90))
    lp = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=4096, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=lerf["bbox"])
    lK = S.lego_K(lh, lw); lc = S.pose_spherical(10.0, -30.0, 4.0)
    l_alone = lerf["renderer"].Render(lh, lw, lK, lp, c2w=lc).Outputs.RenderedLangEmbedding.clone()
    torch.cuda.synchronize()
    together = [None] * len(jobs)
    for rep in range(2):
        for i, (r, h, w, K, rp, c2w) in enumerate(jobs):
            with torch.cuda.stream(streams[i]):
                together[i] = r.Render(h, w, K, rp, c2w=c2w).Outputs.RGBMap
        with torch.cuda.stream(streams[4]):
            l_tog = lerf["renderer"].Render(lh, lw, lK, lp, c2w=lc).Outputs.RenderedLangEmbedding
        with torch.cuda.stream(streams[5]):
            n = 4096; g = torch.Generator().manual_seed(rd)
            o, d, _ = R.GetRays(800, 800, S.lego_K(800, 800), S.pose_spherical(30.0, -30.0, 4.0))
            idx = torch.randint(0, 640000, (n,), generator=g).cuda()
            tr.step(o.reshape(-1, 3)[idx].contiguous(), d.reshape(-1, 3)[idx].contiguous(), torch.rand((n, 3), generator=g).cuda(),
                    R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=n, Perturb=0.0, WhiteBkgr=False, Ndc=False, UseViewdirs=True, ThinRay=True, BoundingBox=S.LEGO_BBOX, Precision=L.NRF_PREC_F16_SPLIT))
        torch.cuda.synchronize()
        ok = all(torch.equal(a, b) for a, b in zip(alone, together)) and torch.equal(l_alone, l_tog) and bool(torch.isfinite(tr.blob).all())
        bad += not ok
        print(f"round {rd} rep {rep}: {'ok' if ok else 'FAIL ' + str([bool(torch.equal(a, b)) for a, b in zip(alone, together)]) + ' lerf ' + str(bool(torch.equal(l_alone, l_tog)))}", flush=True)
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
