"""The small training / producer / post-processing entries against the CPU oracle at random sizes: ray batches (pixel draws, precrop window, gathered colours), huber loss,
RawToOutputs backward (with and without sigma noise), NeRFSmall backward (fp32), both hash-grid backwards, Adam, depth normalisation and 8-bit quantisation.
usage (GPU box): python tools/scratch/train_ops_fuzz.py [cases]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from nerfpp_amd import _lib as L, scene as S, renderer as R, modules as M, dataset as D, synth
from oracle import capi as O
rng = np.random.default_rng(8086)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = L.lib()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
host = lambda t: t.detach().cpu().numpy()
P = lambda t: C.c_void_p(t.data_ptr())
bad = 0
def check(name, got, ref, detail, rtol=0.0, atol=0.0):
    global bad
    got = np.asarray(got); ref = np.asarray(ref).reshape(got.shape)
    ok = np.array_equal(got, ref, equal_nan=True) if (rtol == 0 and atol == 0) else np.allclose(got, ref, rtol=rtol, atol=atol)
    if not ok:
        bad += 1
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        print(f"FAIL {name} {detail}: {int((got != ref).sum())} of {got.size} differ, max {np.nanmax(d):.3e} (scale {np.abs(ref).max():.3e})", flush=True)
def guard(name, detail, fn):
    global bad
    try: fn()
    except Exception as e:
        bad += 1; print(f"FAIL {name} {detail}: EXCEPTION {type(e).__name__}: {str(e)[:240]}", flush=True)
bbox = np.asarray(S.LEGO_BBOX, np.float32)
for i in range(N):
    # ---- ray batches
    h = int(rng.integers(2, 300)); w = int(rng.integers(2, 300)); bs = int(rng.choice([1, 63, 500, 4097])); it = int(rng.integers(0, 1000)); pre = int(rng.choice([0, 500])); seed = int(rng.integers(0, 1 << 30))
    def f():
        img = rng.random((h, w, 3)).astype(np.float32)
        K = S.lego_K(h, w); c2w = S.pose_spherical(float(rng.uniform(-180, 180)), -30.0, 4.0)
        ds = D.NeRFDataset([D.View(H=h, W=w, K=K, Pose=c2w, Image=dev(img), Near=2.0, Far=6.0)], batch_size=bs, precorp_iters=pre, precorp_frac=0.5, seed=seed)
        ds.SetCurrentIter(it)
        b = ds.get_batch()
        bounds = O.precrop_bounds(h, w, it, pre, 0.5)
        check("CalculateBounds", np.array(D.CalculateBounds(h, w, it, pre, 0.5)), np.array(bounds), f"{h}x{w} iter {it}")
        rh, rw = O.rand_pixels(seed, it, bounds, bs)
        check("pixel rows", host(b["rand_h"]), rh, f"{h}x{w} batch {bs}"); check("pixel cols", host(b["rand_w"]), rw, f"{h}x{w} batch {bs}")
        oo, od, cone = O.ray_batch(K, c2w, rh, rw)
        check("batch rays_o", host(b["rays_o"]), oo, f"{h}x{w}"); check("batch rays_d", host(b["rays_d"]), od, f"{h}x{w}")
        check("target colours", host(b["target_s"]), O.gather_pixels(img, rh, rw), f"{h}x{w}")
    guard("ray batch", f"{h}x{w} batch {bs} iter {it} precrop {pre}", f)
    # ---- huber + RawToOutputs backward
    n = int(rng.choice([1, 7, 64, 1000])); s = int(rng.choice([1, 2, 64, 65, 192, 256])); white = bool(rng.integers(0, 2)); nstd = float(rng.choice([0.0, 0.5]))
    def f():
        raw = (rng.standard_normal((n, s, 4)) * 2).astype(np.float32); z = np.sort(rng.uniform(2, 6, (n, s)).astype(np.float32), axis=1); d = rng.standard_normal((n, 3)).astype(np.float32)
        pred = rng.random((n, 3)).astype(np.float32); tgt = rng.random((n, 3)).astype(np.float32)
        lm = torch.empty(2, device="cuda"); g = torch.empty((n, 3), device="cuda")
        d_pred, d_tgt = dev(pred), dev(tgt)               # (named: a temporary handed over as a bare pointer is freed -- and its block reused -- before the call runs)
        L.check(lib.nrf_huber_loss(P(d_pred), P(d_tgt), C.c_int64(n * 3), P(lm), P(g), None))
        ol, om, og = O.huber_loss(pred, tgt)
        check("huber loss", host(lm), np.array([ol, om], np.float32), f"n {n}", rtol=2e-6, atol=1e-9); check("d huber", host(g), og, f"n {n}")
        g_rgb = (rng.standard_normal((n, 3)) * 0.1).astype(np.float32)
        noise = rng.standard_normal((n, s)).astype(np.float32) if nstd > 0 else None
        g_raw = torch.empty((n, s, 4), device="cuda")
        d_raw, d_z, d_d, d_g = dev(raw), dev(z), dev(d), dev(g_rgb); d_noise = dev(noise) if noise is not None else None
        L.check(lib.nrf_raw2outputs_backward_noise(P(d_raw), P(d_z), P(d_d), 3, C.c_int64(n), s, 4, int(white), P(d_noise) if d_noise is not None else None,
                                                   C.c_float(nstd), P(d_g), P(g_raw), None))
        ref = O.raw2outputs_backward_noise(raw, z, d, g_rgb, noise, nstd, white_bkgr=white) if nstd > 0 else O.raw2outputs_backward(raw, z, d, g_rgb, white_bkgr=white)
        check("RawToOutputs backward", host(g_raw), ref, f"n {n} s {s} white {white} noise {nstd}", rtol=2e-5, atol=1e-6 * max(1e-30, float(np.abs(ref).max())))
    guard("loss / compositing backward", f"n {n} s {s}", f)
    # ---- NeRFSmall backward (fp32) and hash backwards
    p = int(rng.choice([1, 63, 65, 1000])); nl = int(rng.choice([2, 3])); nlc = int(rng.choice([2, 3, 4]))
    def f():
        params = S.synth_linear_stack(S.small_shapes(32, 16, nl, 64, 15, nlc, 64), 300 + i, 1.6)
        blob = np.concatenate([a.reshape(-1) for _, a in params])
        m = M.NeRFSmall(nl, 64, 15, nlc, 64, False, 3, 64, 32, 16, "model", params=blob)
        x = rng.uniform(-1, 1, (p, 48)).astype(np.float32); go = (rng.standard_normal((p, 4)) * 0.01).astype(np.float32)
        nb = lib.nrf_mlp_backward_workspace_bytes(m._m, C.c_int64(p)); ws = torch.empty(int(nb), dtype=torch.uint8, device="cuda")
        gp = torch.zeros(m.n_params, device="cuda"); gx = torch.zeros((p, 32), device="cuda")
        d_x, d_go = dev(x), dev(go)
        L.check(lib.nrf_mlp_backward(m._m, P(d_x), P(d_go), C.c_int64(p), P(gp), P(gx), P(ws), C.c_size_t(int(nb)), None))
        rp_, rx_ = O.mlp_small_backward(blob, x, go, 32, 16, nl, 64, 15, nlc, 64)          # (g_params, g_x)
        check("NeRFSmall backward dX", host(gx), rx_, f"p {p} {nl}+{nlc}", rtol=1e-4, atol=1e-6 * float(np.abs(rx_).max()))
        check("NeRFSmall backward dW", host(gp), rp_, f"p {p} {nl}+{nlc}", rtol=1e-3, atol=2e-6 * float(np.abs(rp_).max()))
    guard("NeRFSmall backward", f"p {p} {nl}+{nlc}", f)
    Lv = int(rng.choice([2, 5, 16])); F = int(rng.choice([1, 2, 4])); T = int(rng.choice([10, 14]))
    def f():
        x = rng.uniform(-1.6, 1.6, (p, 3)).astype(np.float32); ge = (rng.standard_normal((p, Lv * F)) * 0.1).astype(np.float32)
        table = synth.synth_sym(9, (Lv * (1 << T) * F,), np.float32(0.5))
        e = M.HashEmbedder("h", bbox, Lv, F, T, 16, 512); e.set_table(table)
        gt = torch.zeros(e.table_elems(), device="cuda")
        d_x, d_ge = dev(x), dev(ge)
        L.check(lib.nrf_hash_backward(e._h, P(d_x), C.c_int64(p), P(d_ge), P(gt), None))
        ref = O.hash_ngp_backward(x, bbox, Lv, F, T, 16, 512, ge)
        check("HashEmbedder backward", host(gt), ref, f"p {p} L {Lv} F {F} T {T}", rtol=1e-4, atol=2e-6 * float(np.abs(ref).max()))
    guard("hash backward", f"p {p} L {Lv} F {F}", f)
    # ---- Adam, post-processing
    n = int(rng.choice([1, 255, 100000]))
    def f():
        prm = rng.standard_normal(n).astype(np.float32); g = rng.standard_normal(n).astype(np.float32) * 0.1; mm = rng.standard_normal(n).astype(np.float32) * 0.01; vv = (rng.random(n) * 1e-3).astype(np.float32)
        t = int(rng.integers(1, 5000)); lr = float(rng.choice([5e-4, 1e-2]))
        dp, dm, dv, dg = dev(prm), dev(mm), dev(vv), dev(g)
        L.check(lib.nrf_adam_step(P(dp), P(dg), P(dm), P(dv), C.c_int64(n), C.c_float(lr), C.c_float(0.9), C.c_float(0.99), C.c_float(1e-15), t, None))
        rp_, rm_, rv_ = prm.copy(), mm.copy(), vv.copy()
        O.adam_step(rp_, g, rm_, rv_, lr, t)                # in place
        check("Adam params", host(dp), rp_, f"n {n} t {t}", rtol=2e-6, atol=1e-9); check("Adam m", host(dm), rm_, f"n {n}", rtol=1e-6, atol=1e-12); check("Adam v", host(dv), rv_, f"n {n}", rtol=1e-6, atol=1e-15)
        dep = rng.uniform(1, 7, (int(rng.integers(1, 50)), int(rng.integers(1, 50)))).astype(np.float32)
        check("NormalizeDepth", host(R.NormalizeDepth(dev(dep), 2.0, 6.0)), O.normalize_depth(dep, 2.0, 6.0), "")
        img = rng.uniform(-0.2, 1.2, dep.shape + (3,)).astype(np.float32)
        check("TorchTensorToCVMat", host(R.TorchTensorToCVMat(dev(img))), O.to_u8(img), "")
    guard("Adam / post", f"n {n}", f)
print("FAILED" if bad else "all equal", bad)
sys.exit(1 if bad else 0)
