"""The stage-level entries against the CPU oracle at RANDOM (also degenerate) sizes, bit for bit: SamplePDF (bins 2.., samples 1..256), RawToOutputs (1..256 samples, both
backgrounds), GetRays tiles, NDCRays, IntersectWithAABB, PE, both SH variants (degree 1..8 / 1..5), both hash encoders (F 1/2/4/8, levels 1..16, tables 2^10..2^16),
the three networks in NRF_PREC_F32, empty batches.  usage (GPU box): python tools/scratch/stage_fuzz.py [cases per stage]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R, modules as M, synth
from oracle import capi as O
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2718)          # second argument: another seed
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
host = lambda t: t.detach().cpu().numpy()
bad = 0
def check(name, got, ref, detail):
    global bad
    got = np.asarray(got); ref = np.asarray(ref).reshape(got.shape)
    ok = np.array_equal(got, ref, equal_nan=True)
    if not ok:
        bad += 1
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        print(f"FAIL {name} {detail}: {int((got != ref).sum())} of {got.size} differ, max {np.nanmax(d):.3e}", flush=True)
    return ok
def guard(name, detail, fn):
    global bad
    try:
        fn()
    except Exception as e:
        bad += 1
        print(f"FAIL {name} {detail}: EXCEPTION {type(e).__name__}: {str(e)[:200]}", flush=True)

bbox = np.asarray(S.LEGO_BBOX, np.float32)
for i in range(N):
    # ---- SamplePDF
    n = int(rng.choice([1, 3, 64, 257])); nb = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 191, 255])); ns = int(rng.choice([1, 2, 64, 128, 129, 256]))
    bins = np.sort(rng.uniform(2, 6, (n, nb)).astype(np.float32), axis=1); wts = (rng.uniform(0, 1, (n, nb - 1)) ** 4).astype(np.float32)
    if rng.integers(0, 3) == 0: wts[rng.integers(0, n)] = 0
    def f():
        smp, inds = R.SamplePDF(dev(bins), dev(wts), ns, det=True, return_inds=True)
        rs, ri, _ = O.sample_pdf(bins, wts, O.linspace(0, 1, ns))
        check("SamplePDF samples", host(smp), rs, f"n {n} bins {nb} samples {ns}"); check("SamplePDF indices", host(inds), ri, f"n {n} bins {nb} samples {ns}")
    guard("SamplePDF", f"n {n} bins {nb} samples {ns}", f)
    # ---- RawToOutputs
    n = int(rng.choice([1, 5, 64, 300])); s = int(rng.choice([1, 2, 63, 64, 65, 192, 256])); white = bool(rng.integers(0, 2))
    raw = rng.standard_normal((n, s, 4)).astype(np.float32) * 3; z = np.sort(rng.uniform(2, 6, (n, s)).astype(np.float32), axis=1); d = rng.standard_normal((n, 3)).astype(np.float32)
    def f():
        sc0 = scenes["cu"]["renderer"]
        o = sc0.RawToOutputs(dev(raw), None, dev(z), dev(d), 0.0, white)
        ro = O.raw2outputs(raw, z, d, white_bkgr=white)
        for k, t in (("rgb", o.RGBMap), ("disp", o.DispMap), ("acc", o.AccMap), ("weights", o.Weights), ("depth", o.DepthMap)):
            check("RawToOutputs " + k, host(t), ro[k] if isinstance(ro, dict) else ro[("rgb", "disp", "acc", "weights", "depth").index(k)], f"n {n} s {s} white {white}")
    scenes = globals().setdefault("scenes", {})
    if "cu" not in scenes: scenes["cu"] = S.make_hash_scene(mode="cu", log2_t=12)
    guard("RawToOutputs", f"n {n} s {s}", f)
    # ---- rays
    h = int(rng.integers(1, 40)); w = int(rng.integers(1, 40)); row0 = int(rng.integers(0, h)); rows = int(rng.integers(0, h - row0 + 1))
    K = S.lego_K(h, w); c2w = S.pose_spherical(float(rng.uniform(-180, 180)), float(rng.uniform(-80, 0)), float(rng.uniform(2, 5)))
    def f():
        o_, d_, cone = R.GetRays(h, w, K, c2w, row0=row0, rows=rows)
        ro, rd = O.get_rays(h, w, K, c2w, row0=row0, rows=rows)[:2]
        check("GetRays o", host(o_), ro, f"{h}x{w} rows {row0}+{rows}"); check("GetRays d", host(d_), rd, f"{h}x{w} rows {row0}+{rows}")
        if rows > 0:
            no, nd, _ = R.NDCRays(h, w, float(K[0][0]) if np.ndim(K) == 2 else float(np.asarray(K).reshape(-1)[0]), 1.0, o_, d_)
            oo, od = O.ndc_rays(h, w, float(np.asarray(K).reshape(-1)[0]), 1.0, host(o_).reshape(-1, 3), host(d_).reshape(-1, 3))
            check("NDCRays o", host(no).reshape(-1, 3), oo, f"{h}x{w}"); check("NDCRays d", host(nd).reshape(-1, 3), od, f"{h}x{w}")
            nr, fr = R.IntersectWithAABB(o_.reshape(-1, 3), d_.reshape(-1, 3), dev(bbox))
            on, of = O.aabb(host(o_).reshape(-1, 3), host(d_).reshape(-1, 3), bbox)
            check("AABB near", host(nr), on, f"{h}x{w}"); check("AABB far", host(fr), of, f"{h}x{w}")
    guard("rays", f"{h}x{w} rows {row0}+{rows}", f)
    # ---- encoders
    p = int(rng.choice([0, 1, 63, 64, 65, 1000])); x = rng.uniform(-1.7, 1.7, (p, 3)).astype(np.float32)
    nf = int(rng.integers(2, 11))          # one frequency: the reference divides MaxFreq by NumFreqs - 1 = 0 (NeRF.cpp:15): NaN bands there and in the oracle, 2^0 here
    def f():
        if p == 0: return
        e = M.Embedder("e", nf); got, _ = e.forward(dev(x)) if isinstance(e.forward(dev(x)), tuple) else (e.forward(dev(x)), None)
        check("PE", host(got), O.pe(x, nf), f"p {p} freqs {nf}")
    guard("PE", f"p {p} freqs {nf}", f)
    deg = int(rng.integers(1, 9)); dirs = rng.standard_normal((max(p, 1), 3)).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    def f():
        got, _ = M.CuSHEncoder("s", 3, deg).forward(dev(dirs)); check("CuSHEncoder", host(got), O.sh_cu(dirs, deg), f"p {len(dirs)} degree {deg}")
        if deg <= 5:
            got, _ = M.SHEncoder("s", 3, deg).forward(dev(dirs)); check("SHEncoder", host(got), O.sh_libtorch(dirs, deg), f"p {len(dirs)} degree {deg}")
    guard("SH", f"degree {deg}", f)
    Lv = int(rng.choice([2, 3, 5, 16]))          # (one level: the reference's growth factor divides by n_levels - 1 = 0, NeRF.cpp:214; refused loudly here)
    F = int(rng.choice([1, 2, 4, 8])); T = int(rng.choice([10, 13, 16])); base = int(rng.choice([4, 16])); fin = int(rng.choice([32, 512, 1024]))
    def f():
        if p == 0: return
        table = synth.synth_sym(int(rng.integers(1, 1000)), (Lv * (1 << T) * F,), np.float32(0.5))
        e = M.HashEmbedder("h", bbox, Lv, F, T, base, fin); e.set_table(table)
        got, keep = e.forward(dev(x))
        ref, rk = O.hash_ngp(x, table, bbox, Lv, F, T, base, fin)
        check("HashEmbedder", host(got), ref, f"p {p} L {Lv} F {F} T {T} {base}..{fin}"); check("HashEmbedder keep", host(keep).astype(np.uint8), np.asarray(rk).astype(np.uint8), f"p {p}")
        c = M.CuHashEmbedder("c", bbox, Lv, F, T, base, fin); c.set_primes(np.array(S.CU_PRIMES[:3 * Lv], np.int32)); c.set_table(table)
        got, keep = c.forward(dev(x))
        ls = ((1 << T) >> 4) << 4
        ref, rk = O.hash_cu(x, O.f32_to_f16(table), np.array(S.CU_PRIMES[:3 * Lv], np.int32), np.arange(Lv, dtype=np.int32) * ls, np.full(Lv, ls, np.int32), np.zeros((Lv, 3), np.float32), bbox,
                            O.hash_cu_scales(Lv, base, fin), Lv, F)
        check("CuHashEmbedder", host(got), ref, f"p {p} L {Lv} F {F} T {T} {base}..{fin}"); check("CuHashEmbedder keep", host(keep).astype(np.uint8), np.asarray(rk).astype(np.uint8), f"p {p}")
    guard("hash", f"p {p} L {Lv} F {F} T {T}", f)
print("FAILED" if bad else "all equal", bad)
sys.exit(1 if bad else 0)
