"""The LeRF frame at RANDOM sizes / sample counts (multiples of 32) / chunk sizes / tiles / precisions: Chunk-invariance, tile == rows of the frame, single library call ==
the stage-wise host loop, finite outputs (also for rays that miss the box), unit-norm embeddings, relevancy in [0, 1].  usage (GPU box): python tools/scratch/lerf_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)          # second argument: another seed
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sc = S.make_lerf_scene(log2_t=14); r = sc["renderer"]
pr = np.random.RandomState(5); pos = pr.randn(1, 768).astype(np.float32); pos /= np.linalg.norm(pos); neg = pr.randn(3, 768).astype(np.float32); neg /= np.linalg.norm(neg, axis=1, keepdims=True)
r.SetLeRFPrompts(pos, neg)
bad = 0
exact = lambda a, b: a.shape == b.shape and torch.equal(a.nan_to_num(nan=4321.0), b.nan_to_num(nan=4321.0))
# NRF_PREC_F16_MFMA sums a ray's embedding with float atomics (one per 32-sample tile): reproducible to summation order only; the split mode's wave owns its ray
close = lambda a, b: a.shape == b.shape and bool(torch.isfinite(a).all()) and float((a.float() - b.float()).abs().max()) <= 2e-5 * max(1.0, float(b.float().abs().max()))
eq = exact
for case in range(cases):
    h = int(rng.integers(5, 90)); w = int(rng.integers(5, 90)); n = h * w
    s = int(rng.choice([32, 64, 96])); ni = int(rng.choice([32, 64, 128]))
    prec = int(rng.choice([L.NRF_PREC_F16_SPLIT, L.NRF_PREC_F16_MFMA]))
    r.set_precision(prec)
    eq = exact if prec == L.NRF_PREC_F16_SPLIT else close
    K = S.lego_K(h, w); c2w = S.pose_spherical(float(rng.uniform(-180, 180)), float(rng.uniform(-60, -5)), float(rng.uniform(3.0, 4.6)))
    chunks = [n, int(rng.integers(max(1, n // 5), n + 1)), int(rng.choice([33, 1000, 4096]))]
    msgs = []
    try:
        def render(chunk, **extra):
            p = R.NeRFRenderParams(NSamples=s, NImportance=ni, Chunk=chunk, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=sc["bbox"])
            return r.Render(h, w, K, p, c2w=c2w, **extra).Outputs
        full = render(chunks[0])
        W = lambda o: getattr(o, "Weights", None) if getattr(o, "Weights", None) is not None else getattr(o, "WeightsLE", None)
        fo = dict(emb=full.RenderedLangEmbedding, rel=getattr(full, "Relevancy", None), w=W(full), depth=getattr(full, "DepthMapLE", getattr(full, "DepthMap", None)))
        for k, v in fo.items():
            if v is not None and not bool(torch.isfinite(v).all()): msgs.append(f"{k} non-finite")
        nrm = fo["emb"].reshape(-1, 768).norm(dim=1)
        if float((nrm - 1).abs().max()) > 1e-4 and float(nrm.min()) > 1e-6: msgs.append(f"embedding norms {float(nrm.min()):.6f}..{float(nrm.max()):.6f}")
        if fo["rel"] is not None and (float(fo["rel"].min()) < 0 or float(fo["rel"].max()) > 1): msgs.append("relevancy outside [0, 1]")
        for ch in chunks[1:]:
            o = render(ch)
            if not (eq(o.RenderedLangEmbedding, fo["emb"]) and (fo["w"] is None or eq(W(o), fo["w"]))): msgs.append(f"chunk {ch} differs")
        row0 = int(rng.integers(0, h)); rows = int(rng.integers(1, h - row0 + 1))
        t = render(chunks[1], row0=row0, rows=rows)
        if not eq(t.RenderedLangEmbedding.reshape(rows, w, 768), fo["emb"].reshape(h, w, 768)[row0:row0 + rows]): msgs.append(f"tile {row0}+{rows} differs")
        r.single_call = False
        st = render(chunks[1])
        r.single_call = True
        if not eq(st.RenderedLangEmbedding.reshape(fo["emb"].shape), fo["emb"]): msgs.append("stage-wise host loop differs from the single library call")
    except Exception as e:
        r.single_call = True
        msgs.append(f"EXCEPTION {type(e).__name__}: {str(e)[:200]}")
    bad += bool(msgs)
    print(f"case {case:2d}: {h}x{w} s {s}+{ni} precision {prec} chunks {chunks}: {'ok' if not msgs else 'FAIL ' + '; '.join(msgs)}", flush=True)
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
