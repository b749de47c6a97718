"""classic NeRF frame time per precision (tuning builds via NRF_LIB_PATH); args: precisions, e.g. f16x3 f16 f32"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerfpp_amd import _lib as L, scene as S
H = W = 800
sc = S.make_classic_scene()
K = S.lego_K(H, W); c2w = S.pose_spherical(30.0, -30.0, 4.0)
for name in (sys.argv[1:] or ["f16x3", "f16"]):
    prec = {"f16x3": L.NRF_PREC_F16_SPLIT, "f16": L.NRF_PREC_F16_MFMA, "f32": L.NRF_PREC_F32}[name]
    rows = H if name != "f32" else 40
    rp = S.lego_render_params(sc["bbox"], 64, 128, 8192, prec)
    sc["renderer"].Render(H, W, K, rp, c2w=c2w, row0=(H - rows) // 2, rows=rows)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3 if name != "f32" else 1):
        t0 = time.perf_counter(); out = sc["renderer"].Render(H, W, K, rp, c2w=c2w, row0=(H - rows) // 2, rows=rows); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = min(ts) * H / rows
    print(name, "ms/frame %.1f" % (dt * 1e3), "units/s %.3e" % (H * W * 256 / dt), "TFLOP/s algorithmic %.0f" % (H * W * 256 * 1186816 / dt / 1e12), "rgb mean %.6f" % float(out.Outputs.RGBMap.mean()), flush=True)
