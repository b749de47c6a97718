"""NRF_PREC_F32 against the CPU oracle, bit for bit, at RANDOM sample counts / ray counts / chunk sizes / scenes / stochastic settings (the test suite pins fixed sizes):
z_vals, jitter, TangentScatter, preconditioning, the encoders, the network, RawToOutputs with noise, SamplePDF (deterministic and drawn), the merge -- one oracle call per
case on the same packed rays and the same counter-based draws.  usage (GPU box): python tools/scratch/oracle_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R
from oracle import capi as O
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)          # second argument: another seed
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
def models(seed=0):
    out = []
    g = np.random.default_rng(seed)
    for mode in ("cu", "ngp"):
        sc = S.make_hash_scene(mode=mode, log2_t=int(g.choice([12, 14, 16])), base=int(g.choice([4, 16])), finest=int(g.choice([128, 512, 1024])), seed=5000 + seed); cfg = sc["cfg"]
        sc["embedder"].set_dense_budget(int(g.choice([0, 4 << 20, 1 << 30])))
        if mode == "cu":
            ls = ((1 << cfg["log2_t"]) >> 4) << 4; Lv = cfg["n_levels"]
            m = O.Model(2, sc["mlp_blob"], bbox=sc["bbox"], table_f16=O.f32_to_f16(sc["table"]), primes=sc["primes"], local_idx=np.arange(Lv, dtype=np.int32) * ls,
                        local_size=np.full(Lv, ls, np.int32), bias=np.zeros((Lv, 3), np.float32), mul=O.hash_cu_scales(Lv, cfg["base"], cfg["finest"]), log2_t=cfg["log2_t"],
                        base=cfg["base"], finest=cfg["finest"])
        else:
            m = O.Model(0, sc["mlp_blob"], bbox=sc["bbox"], table_f32=sc["table"], log2_t=cfg["log2_t"], base=cfg["base"], finest=cfg["finest"])
        out.append((mode, sc, m))
    sc = S.make_classic_scene()
    out.append(("classic", sc, O.Model(1, sc["mlp_blob"], bbox=sc["bbox"])))
    return out
ms = models()
bad = 0
for case in range(cases):
    if case % 10 == 0 and case: ms = models(case)          # a new grid configuration (table size, resolutions, dense budget) every ten cases
    name, sc, model = ms[int(rng.integers(0, 3))]; r = sc["renderer"]
    h = int(rng.integers(3, 24)); w = int(rng.integers(3, 24))
    if name == "classic": h, w = min(h, 10), min(w, 12)
    s = int(rng.choice([2, 3, 4, 8, 17, 32, 64, 65, 100])); ni = int(rng.choice([0, 1, 5, 32, 63, 128]))
    if s + ni > 256: ni = 256 - s
    stoch = bool(rng.integers(0, 2)); white = bool(rng.integers(0, 2)); lindisp = bool(rng.integers(0, 4) == 0)
    seed = int(rng.integers(1, 1 << 40))
    kw = dict(WhiteBkgr=white, LinDisp=lindisp)
    st = None
    o_, d_, cone = R.GetRays(h, w, S.lego_K(h, w), S.pose_spherical(float(rng.uniform(-180, 180)), float(rng.uniform(-60, -5)), float(rng.uniform(3.2, 4.4))))
    if stoch:
        thin = bool(rng.integers(0, 2)); noise = float(rng.choice([0.0, 0.4])); pre = float(rng.choice([0.0, 0.02]))
        kw.update(Perturb=1.0, ThinRay=thin, Seed=seed, RawNoiseStd=noise, StochasticPreconditioningAlpha=pre)
        st = dict(perturb=1.0, cone_angle=None if thin else float(cone), seed=seed, raw_noise_std=noise, precond_alpha=pre)
    n = h * w
    chunk = int(rng.choice([n, max(1, n // 3), 7, 64]))
    K = S.lego_K(h, w)
    pose = None
    try:
        rp = S.lego_render_params(sc["bbox"], s, ni, chunk, L.NRF_PREC_F32, ReturnWeights=True, KeepIntermediates=True, **{k: v for k, v in kw.items() if k != "WhiteBkgr"}, white_bkgr=white)
        res = r.Render(h, w, K, rp, rays=(o_, d_, cone))
        rays = res.Extras["rays_flat"].cpu().numpy()
        oc = O.render_rays(model, rays, s, ni, O.linspace(0, 1, s), None if (stoch or ni == 0) else O.linspace(0, 1, ni), lindisp=lindisp, white_bkgr=white, want_intermediates=True, stoch=st)
        got = dict(rgb=res.Outputs.RGBMap.reshape(-1, 3), acc=res.Outputs.AccMap.reshape(-1), depth=res.Outputs.DepthMap.reshape(-1), disp=res.Outputs.DispMap.reshape(-1),
                   weights=res.Outputs.Weights.reshape(n, -1), z_coarse=res.Extras["z_coarse"].reshape(n, -1))
        if ni > 0: got["z_fine"] = res.Extras["z_fine"].reshape(n, -1)
        msgs = []
        for k, v in got.items():
            a = v.cpu().numpy(); b = oc[k].reshape(a.shape)
            if not np.array_equal(a, b, equal_nan=True):
                d = np.abs(a.astype(np.float64) - b.astype(np.float64)); msgs.append(f"{k}: {int((a != b).sum())} of {a.size} differ, max {np.nanmax(d):.3e}")
        ok = not msgs
    except Exception as e:
        ok = False; msgs = [f"EXCEPTION {type(e).__name__}: {str(e)[:200]}"]
    bad += not ok
    print(f"case {case:2d}: {name} {h}x{w} s {s}+{ni} chunk {chunk} white {white} lindisp {lindisp} stoch {st}: {'== oracle' if ok else 'FAIL ' + '; '.join(msgs)}", flush=True)
print("FAILED" if bad else "all equal", bad)
sys.exit(1 if bad else 0)
