"""Random frame sizes x chunk sizes x lane counts: the Chunk loop on L lanes must reproduce the single-stream loop bit for bit (every output), whatever the
remainders of the lane scheduler (stagger, balance, sliver and crumb rules of nrf_batchify_rays).  usage (GPU box): python tools/scratch/lane_fuzz.py [cases]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)          # second argument: another seed
scenes = [S.make_hash_scene(mode="cu", log2_t=16), S.make_hash_scene(mode="ngp", log2_t=16), S.make_classic_scene()]
lsc = S.make_lerf_scene(log2_t=14); lr = lsc["renderer"]
lib = L.lib()
bad = 0
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for case in range(cases):
    h = int(rng.integers(150, 520)); w = int(rng.integers(150, 520))
    chunk = int(rng.choice([rng.integers(33000, 140000), rng.integers(33000, 70000), 32768, 65536, h * w // 2 + 1, h * w - 1, h * w]))
    K = S.lego_K(h, w); c2w = S.pose_spherical(float(rng.uniform(-180, 180)), float(rng.uniform(-60, -5)), float(rng.uniform(3.3, 4.5)))
    prec = [L.NRF_PREC_F16_SPLIT, L.NRF_PREC_F16_MFMA, L.NRF_PREC_F32][case % 3 if case % 7 else 0]
    which = int(rng.integers(0, 3)); sc = scenes[which]; r = sc["renderer"]
    if which == 2:                                   # the classic network is ~25x the work per sample: a smaller frame
        h, w = h // 3 + 40, w // 3 + 40
        chunk = max(33000 // 9, chunk // 9)
    stoch = bool(rng.integers(0, 3) == 0)
    rp = S.lego_render_params(sc["bbox"], 64, 128, chunk, prec, ReturnWeights=True, **(dict(Perturb=1.0, ThinRay=False, Seed=int(rng.integers(1, 1 << 30))) if stoch else {}))
    K = S.lego_K(h, w)
    outs = []
    for lanes in (1, 2, 3, 4):
        L.check(lib.nrf_set_render_lanes(lanes))
        o = r.Render(h, w, K, rp, c2w=c2w).Outputs
        outs.append([t.clone() for t in (o.RGBMap, o.DepthMap, o.AccMap, o.DispMap, o.Weights)])
    torch.cuda.synchronize()
    ok = all(torch.equal(a.nan_to_num(nan=12345.0), b.nan_to_num(nan=12345.0)) for k in range(1, 4) for a, b in zip(outs[0], outs[k]))
    fin = all(bool(torch.isfinite(t).all()) for t in outs[0][:3])
    bad += (not ok) or (not fin)
    print(f"case {case:2d}: scene {('cu', 'ngp', 'classic')[which]}{' stochastic' if stoch else ''} {h}x{w} = {h * w} rays, chunk {chunk}, precision {prec}: lanes 2-4 == 1: {ok}, finite {fin}", flush=True)
L.check(lib.nrf_set_render_lanes(2))
# the LeRF frame: its own lane count per renderer (1-4), same statement
for case in range(max(4, cases // 5)):
    h = int(rng.integers(60, 160)); w = int(rng.integers(60, 160))
    chunk = int(rng.choice([rng.integers(1000, 9000), 4096, h * w // 3 + 7]))
    K = S.lego_K(h, w); c2w = S.pose_spherical(float(rng.uniform(-180, 180)), -30.0, 4.0)
    from nerfpp_amd import renderer as R
    p = R.NeRFRenderParams(NSamples=64, NImportance=128, Chunk=chunk, Perturb=0.0, Ndc=False, UseViewdirs=True, ReturnWeights=True, ThinRay=True, BoundingBox=lsc["bbox"])
    embs = []
    for lanes in (1, 2, 4):
        lr.lanes = lanes
        if hasattr(lr, "_r") and lr._r: L.check(lib.nrf_lerf_renderer_set_lanes(lr._r, lanes))
        embs.append(lr.Render(h, w, K, p, c2w=c2w).Outputs.RenderedLangEmbedding.clone())
    torch.cuda.synchronize()
    ok = all(torch.equal(embs[0], e) for e in embs[1:])
    bad += not ok
    print(f"lerf case {case}: {h}x{w}, chunk {chunk}: lanes 2, 4 == 1: {ok}", flush=True)
print("FAILED" if bad else "all equal", bad)
sys.exit(1 if bad else 0)
