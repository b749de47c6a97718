"""Random sample counts / frame sizes / tiles / chunk sizes over the three scenes and precisions: a render must not depend on Chunk, a row tile must equal the rows of the
whole frame, the explicit ray batch must equal the pose branch, everything finite -- bit for bit, the stochastic branches included (draws are keyed by the ray's index in the
frame).  usage (GPU box): python tools/scratch/render_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)          # second argument: another seed
scenes = [("cu", S.make_hash_scene(mode="cu", log2_t=15)), ("ngp", S.make_hash_scene(mode="ngp", log2_t=15)), ("classic", S.make_classic_scene())]
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
eq = lambda a, b: a.shape == b.shape and torch.equal(a.nan_to_num(nan=4321.0), b.nan_to_num(nan=4321.0))
for case in range(cases):
    name, sc = scenes[int(rng.integers(0, 3))]; r = sc["renderer"]
    h = int(rng.integers(9, 70)); w = int(rng.integers(9, 70))
    if name != "classic": h *= 2; w *= 2
    s = int(rng.choice([2, 3, 8, 17, 32, 64, 100])); ni = int(rng.choice([0, 5, 32, 128, 150]))
    if s + ni > 256: ni = 256 - s
    prec = int(rng.choice([L.NRF_PREC_F32, L.NRF_PREC_F16_MFMA, L.NRF_PREC_F16_SPLIT]))
    stoch = bool(rng.integers(0, 3) == 0) and s >= 2
    kw = dict(Perturb=1.0, ThinRay=False, Seed=int(rng.integers(1, 1 << 30)), RawNoiseStd=float(rng.choice([0.0, 0.3]))) if stoch else {}
    n = h * w
    c_all = max(n, 1); c_a = int(rng.integers(max(1, n // 7), n + 1)); c_b = int(rng.choice([64, 1000, 4097, n - 1 if n > 1 else 1]))
    K = S.lego_K(h, w); c2w = S.pose_spherical(float(rng.uniform(-180, 180)), float(rng.uniform(-60, -5)), float(rng.uniform(3.0, 4.6)))
    def render(chunk, **extra):
        rp = S.lego_render_params(sc["bbox"], s, ni, chunk, prec, ReturnWeights=True, **kw)
        return r.Render(h, w, K, rp, c2w=c2w, **extra)
    try:
        full = render(c_all)
        fo = [full.Outputs.RGBMap, full.Outputs.DepthMap, full.Outputs.AccMap, full.Outputs.DispMap, full.Outputs.Weights]
        ok = all(bool(torch.isfinite(t).all()) for t in fo)
        msgs = [] if ok else ["non-finite"]
        for ch in (c_a, c_b):
            o = render(ch).Outputs
            if not all(eq(a, b) for a, b in zip(fo, [o.RGBMap, o.DepthMap, o.AccMap, o.DispMap, o.Weights])): ok = False; msgs.append(f"chunk {ch} differs")
        row0 = int(rng.integers(0, h)); rows = int(rng.integers(1, h - row0 + 1))
        t = render(c_a, row0=row0, rows=rows).Outputs
        tt = [t.RGBMap, t.DepthMap, t.Weights]; ff = [fo[0].reshape(h, w, -1)[row0:row0 + rows], fo[1].reshape(h, w)[row0:row0 + rows], fo[4].reshape(h, w, -1)[row0:row0 + rows]]
        if not all(eq(a.reshape(b.shape), b) for a, b in zip(tt, ff)): ok = False; msgs.append(f"tile {row0}+{rows} differs (shapes {[tuple(a.shape) for a in tt]} vs {[tuple(b.shape) for b in ff]})")
        if not stoch:                                    # the ray-batch branch has no frame position to key draws by
            o_, d_, cone = R.GetRays(h, w, K, c2w)
            rp = S.lego_render_params(sc["bbox"], s, ni, c_a, prec, ReturnWeights=True)
            b = r.Render(h, w, K, rp, rays=(o_, d_, cone)).Outputs
            if not (eq(b.RGBMap.reshape(fo[0].shape), fo[0]) and eq(b.Weights.reshape(fo[4].shape), fo[4])): ok = False; msgs.append("ray batch differs")
    except Exception as e:
        ok = False; msgs = [f"EXCEPTION {type(e).__name__}: {str(e)[:160]}"]
    bad += not ok
    print(f"case {case:2d}: {name}{' stochastic' if stoch else ''} {h}x{w} s {s}+{ni} precision {prec} chunks {c_all}/{c_a}/{c_b}: {'ok' if ok else 'FAIL ' + '; '.join(msgs)}", flush=True)
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
