#!/usr/bin/env python3
"""Per-layer product budget of the split-precision NeRFSmall kernel (k_mlp_small_mfma<SPLIT>): which of the three fp16 products per layer does the frame need?

Every fp32 quantity travels as hi + lo fp16 halves and a layer's W.x is Wh.xh + Wl.xh + Wh.xl (mlp_small_mfma.hip).  Libraries built with NRF_SMALL_DROP_MASK
(tools/product_budget_build.sh: bit 2 id drops Wl.xh of layer id, bit 2 id + 1 drops Wh.xl and the forming of the lo halves it would read; ids 0-2 sigma net, 3-6 colour
net) render the bench frame (HashNeRF, 800x800, 64 + 128) and are compared, whole frame, with NRF_PREC_F32 of the shipped library (== the CPU oracle bit for bit).

    run on the GPU box:   python tools/product_budget.py 0x0 0x4 0x8 ...        one JSON line per mask + a table on stderr
Reference semantics: NeRFSmallImpl::forward, NeRF.cpp:363-408."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = W = 800


def describe(mask):
    names = ["s0", "s1", "s2", "c0", "c1", "c2", "c3"]
    out = []
    for i, n in enumerate(names):
        b = (mask >> (2 * i)) & 3
        if b:
            out.append(n + ":" + {1: "-Wl", 2: "-xl", 3: "-Wl-xl"}[b])
    return " ".join(out) or "all products"


def issued(mask):
    """matrix instructions per 32 points: whole network / colour net alone"""
    steps = [(2, 2, False), (4, 2, True), (4, 1, True), (2, 2, True), (4, 2, True), (4, 2, True), (4, 1, True)]      # (k-steps, m-tiles, operand has a lo half)
    tot = []
    for i, (ks, mt, blo) in enumerate(steps):
        b = (mask >> (2 * i)) & 3
        n = 1 + (0 if b & 1 else 1) + ((0 if b & 2 else 1) if blo else 0)
        tot.append(ks * mt * n)
    return sum(tot), sum(tot[3:])


def worker(mask_s, ref_path):
    import ctypes as C
    import numpy as np
    import torch
    from nerfpp_amd import _lib as L, scene
    sc = scene.make_hash_scene(mode="cu")
    K = scene.lego_K(H, W); c2w = scene.pose_spherical(30.0, -30.0, 4.0)
    r = sc["renderer"]
    if mask_s == "ref":
        rp = scene.lego_render_params(sc["bbox"], 64, 128, 32768, L.NRF_PREC_F32)
        np.save(ref_path, r.Render(H, W, K, rp, c2w=c2w).Outputs.RGBMap.cpu().numpy())
        return
    ref = np.load(ref_path)
    rp = scene.lego_render_params(sc["bbox"], 64, 128, 131072, L.NRF_PREC_F16_SPLIT)
    img = r.Render(H, W, K, rp, c2w=c2w).Outputs.RGBMap
    torch.cuda.synchronize()
    d = np.abs(img.cpu().numpy() - ref)
    import time
    for _ in range(3):
        r.Render(H, W, K, rp, c2w=c2w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        r.Render(H, W, K, rp, c2w=c2w)
    torch.cuda.synchronize()
    frame_ms = (time.perf_counter() - t0) / 10 * 1e3
    n = len(L.NRF_PROF_NAMES)
    ms = (C.c_double * n)(); cnt = (C.c_int64 * n)()
    L.lib().nrf_set_render_lanes(1)
    r.Render(H, W, K, rp, c2w=c2w); torch.cuda.synchronize()
    L.lib().nrf_profile_enable(1); L.lib().nrf_profile_read(ms, cnt, 1)
    for _ in range(5):
        r.Render(H, W, K, rp, c2w=c2w)
    torch.cuda.synchronize()
    L.lib().nrf_profile_read(ms, cnt, 1); L.lib().nrf_profile_enable(0)
    k = {nm: ms[i] / 5 for i, nm in enumerate(L.NRF_PROF_NAMES)}
    mask = int(mask_s, 16)
    mse = float((d.astype(np.float64) ** 2).mean())
    print(json.dumps(dict(mask=mask_s, drops=describe(mask), mfma_per_32_points=issued(mask)[0], colour_only_mfma=issued(mask)[1], max_abs_err=float(d.max()),
                          frac_gt_1e5=float((d > 1e-5).mean()), frac_gt_1e4=float((d > 1e-4).mean()), psnr_db=(-10 * np.log10(mse) if mse > 0 else 999.0),
                          mlp_ms=k["mlp"], mlp_colour_ms=k["mlp_colour"], frame_ms_two_lanes=frame_ms)), flush=True)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--worker":
        return worker(sys.argv[2], sys.argv[3])
    masks = sys.argv[1:] or ["0x0"]
    ref = "/tmp/pb_ref.npy"
    subprocess.check_call([sys.executable, __file__, "--worker", "ref", ref])
    rows = []
    for m in masks:
        lib = os.path.join(ROOT, "tune", f"pb_{m}", "libnerfpp_hip.so")
        if not os.path.exists(lib):
            print(f"[skip] {lib} not built", file=sys.stderr)
            continue
        out = subprocess.run([sys.executable, __file__, "--worker", m, ref], capture_output=True, text=True, env=dict(os.environ, NRF_LIB_PATH=lib), timeout=600)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(f"[fail] {m}: {out.stderr[-400:]}", file=sys.stderr)
            continue
        print(line[-1], flush=True)
        rows.append(json.loads(line[-1]))
    print(f"{'mask':>8} {'drops':<28} {'mfma':>5} {'max err':>10} {'>1e-5':>8} {'>1e-4':>8} {'psnr':>7} {'mlp ms':>7} {'col ms':>7} {'frame':>7}", file=sys.stderr)
    for r in rows:
        print(f"{r['mask']:>8} {r['drops']:<28} {r['mfma_per_32_points']:>5} {r['max_abs_err']:>10.2e} {r['frac_gt_1e5']:>8.4f} {r['frac_gt_1e4']:>8.5f} {r['psnr_db']:>7.1f} "
              f"{r['mlp_ms']:>7.2f} {r['mlp_colour_ms']:>7.2f} {r['frame_ms_two_lanes']:>7.2f}", file=sys.stderr)


if __name__ == "__main__":
    main()
